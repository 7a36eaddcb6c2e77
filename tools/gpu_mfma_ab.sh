#!/bin/bash
# Round 6, Poseidon on the matrix pipe: parity of each variant library on the Poseidon / LinearHash / Merkle tests, then A/B timings on one box.
# gpurun -- 'bash tools/gpu_mfma_ab.sh mf7 ...'   ("shipped" is always run; name@ENV=VAL runs a variant under an environment switch)
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/mfma_ab.txt; : > $out
sel() { v=${1%%@*}; e=""; [ "$1" != "$v" ] && e=${1#*@}; if [ $v = shipped ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; fi; }
for a in shipped "$@"; do
  sel $a
  echo "== parity $a" >> $out
  env $e timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 >> $out
done
for r in 1 2; do
  for a in shipped "$@"; do
    sel $a
    echo "== $a (run $r)" >> $out
    env $e timeout 300 python tools/merkle_bench.py 22 19 22 36 18 12 16 37 >> $out 2>&1
  done
done
unset ZKGPU_LIB
cat $out
