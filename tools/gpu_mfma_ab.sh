#!/bin/bash
# Poseidon-GL variants on one box: parity of each variant library on the Poseidon / LinearHash / Merkle tests, then A/B tree timings.
# gpurun -- 'bash tools/gpu_mfma_ab.sh name[@ENV=VAL] ...'   (variants of eigen-zkvm_amd/variants/ built by tools/build_variant.sh; "shipped" = the library)
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/mfma_ab.txt; : > $out
sel() { v=${1%%@*}; e=""; [ "$1" != "$v" ] && e=${1#*@}; if [ $v = shipped ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; fi; }
for a in "$@"; do
  sel $a
  echo "== parity $a" >> $out
  env $e timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 >> $out
done
for r in 1 2; do
  for a in "$@"; do
    sel $a
    echo "== $a (run $r)" >> $out
    env $e timeout 300 python tools/merkle_bench.py 22 19 22 36 18 12 16 37 24 10 2>&1 | cut -c1-50 >> $out
  done
done
unset ZKGPU_LIB
cat $out
