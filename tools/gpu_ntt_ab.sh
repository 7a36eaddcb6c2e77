#!/bin/bash
# NTT variants on one box: gpurun -- 'bash tools/gpu_ntt_ab.sh name ...' (variants of eigen-zkvm_amd/variants/; "shipped" = the library): the NTT / LDE parity
# tests under each library, then 2^24 x 1 forward + inverse and the 2^24 -> 2^25 extensions of 19 / 36 columns, alternating
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/ntt_ab.txt; : > $out
sel() { if [ $1 = shipped ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$1.so; fi; }
for a in "$@"; do
  sel $a; echo "== parity $a" >> $out
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lde or fft or interpolate or root" 2>&1 | tail -2 >> $out
done
for r in 1 2; do
  for a in "$@"; do
    sel $a; echo "== $a (run $r)" >> $out
    for sh in "24 1" "20 36"; do timeout 300 python tools/ntt_time.py $sh >> $out 2>&1; done
    timeout 300 python tools/lde_time.py 24 36 >> $out 2>&1
  done
done
unset ZKGPU_LIB
cat $out
