#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" lh3 lh5; do
  if [ -n "$v" ]; then export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; fi
  echo "== variant '$v'"; timeout 300 python tools/merkle_bench.py 22 19 22 36 22 6
done > gpurun_out/c_variants.log 2>&1
unset ZKGPU_LIB
cat gpurun_out/c_variants.log
bash tools/gpu_pmc_sq.sh > gpurun_out/c_pmc_sq.txt 2>&1; grep "linearhash\|merkle_level_kernel" gpurun_out/c_pmc_sq.txt | head -4
