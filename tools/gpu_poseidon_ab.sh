#!/bin/bash
# A/B of Poseidon-GL builds on one box: gpurun -- 'bash tools/gpu_poseidon_ab.sh v1 v2 ...' (variants of eigen-zkvm_amd/variants/, "shipped" = the library)
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/poseidon_ab.txt; : > $out
for r in 1 2; do
  for v in "$@"; do
    if [ $v = shipped ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; fi
    echo "== $v (run $r)" >> $out
    timeout 300 python tools/merkle_bench.py 22 19 22 36 >> $out 2>&1
    timeout 100 python tools/coop_perm_time.py 2>&1 | tail -1 >> $out
  done
done
unset ZKGPU_LIB
cat $out
