#!/bin/bash
# timings only: gpurun -- 'bash tools/gpu_mfma_ab2.sh name[@ENV=VAL] ...'
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/mfma_ab2.txt; : > $out
sel() { v=${1%%@*}; e=""; [ "$1" != "$v" ] && e=${1#*@}; if [ $v = shipped ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; fi; }
for r in 1 2; do
  for a in "$@"; do
    sel $a
    echo "== $a (run $r)" >> $out
    env $e timeout 300 python tools/merkle_bench.py 22 19 22 36 18 12 16 37 2>&1 | cut -c1-48 >> $out
  done
done
unset ZKGPU_LIB
cat $out
