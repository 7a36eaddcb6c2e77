#!/bin/bash
# the register leaf kernels for 9..16 blocks against the generic one: gpurun -- 'bash tools/gpu_fr_reg.sh'
mkdir -p gpurun_out; : > gpurun_out/fr_reg_raw.txt
for w in 27 30 33 36 37 39 42 45 48; do
  for m in 8 16; do
    echo "ZK_FRHASH_REG_MAX=$m width $w" >> gpurun_out/fr_reg_raw.txt
    ZK_FRHASH_REG_MAX=$m timeout 300 python3 tools/fr_merkle_time.py bls12381 17 $w 3 >> gpurun_out/fr_reg_raw.txt 2>&1
  done
done
for lv in 16384 4096 1024 256; do
  echo "ZK_FR_LEVEL_COOP=$lv" >> gpurun_out/fr_reg_raw.txt
  ZK_FR_LEVEL_COOP=$lv timeout 300 python3 tools/fr_merkle_time.py bls12381 17 12 3 >> gpurun_out/fr_reg_raw.txt 2>&1
done
cat gpurun_out/fr_reg_raw.txt | cut -c1-120
