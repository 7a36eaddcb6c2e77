#!/bin/bash
# tree levels with eight lanes per permutation against one wave each / one lane each: gpurun -- 'bash tools/gpu_fr_grp.sh'
mkdir -p gpurun_out; out=gpurun_out/fr_grp_raw.txt; : > $out
for lg in 13 15 17 18 19; do
  for cfg in "ZK_FR_LEVEL_GRP_UPTO=0" "ZK_FR_LEVEL_GRP_FROM=1025 ZK_FR_LEVEL_GRP_UPTO=32768" "ZK_FR_LEVEL_GRP_FROM=513 ZK_FR_LEVEL_GRP_UPTO=65536"; do
    echo "== $cfg" >> $out
    env $cfg timeout 300 python3 tools/fr_merkle_time.py bls12381 $lg 3 5 2>&1 | cut -c1-60 >> $out
  done
done
cat $out
