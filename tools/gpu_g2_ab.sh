#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/g2_ab.txt; : > $out
for r in 1 2; do
  for v in shipped g2w2; do
    if [ $v = shipped ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; fi
    echo "== $v (run $r)" >> $out
    timeout 300 python tools/msm_bench.py bn254 g2 20 22 2>&1 | grep msm >> $out
    timeout 300 python tools/msm_bench.py bls12_381 g2 20 2>&1 | grep msm >> $out
    timeout 300 python tools/groth16_bench.py BN128 20 2>&1 | tail -2 >> $out
  done
done
unset ZKGPU_LIB
cat $out
