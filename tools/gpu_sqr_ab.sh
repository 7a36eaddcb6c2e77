#!/bin/bash
# dedicated squaring (fe_sqr) against fe_mul(a, a) in the MSM: gpurun -- 'bash tools/gpu_sqr_ab.sh'   (variant: tools/build_variant.sh sqrplain msm.hip -DZK_FE_SQR_PLAIN)
mkdir -p gpurun_out; out=gpurun_out/sqr_ab_raw.txt; : > $out
V=eigen-zkvm_amd/variants/libzkgpu_sqrplain.so
for r in 1 2; do
  for c in "bn254 g1 22" "bls12_381 g1 22" "bn254 g2 20" "bls12_381 g2 20"; do
    echo "== fe_sqr dedicated: $c (run $r)" >> $out; timeout 300 python3 tools/msm_bench.py $c 2>&1 | grep "^msm" >> $out
    echo "== fe_mul(a, a): $c (run $r)" >> $out; ZKGPU_LIB=$V timeout 300 python3 tools/msm_bench.py $c 2>&1 | grep "^msm" >> $out
  done
done
cat $out
