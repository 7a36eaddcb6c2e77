#!/bin/bash
# row-hash kernel switch points on the 2^18-row circuit of a recursion task (its first FRI tree: 2^13 rows of 192 words)
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/lh_thresholds.txt; : > $out
for r in 1 2; do
  for v in default 8192 4097; do
    if [ $v = default ]; then unset ZK_LH_COOP_BELOW; else export ZK_LH_COOP_BELOW=$v; fi
    echo "== ZK_LH_COOP_BELOW=$v (run $r)" >> $out
    for k in r1 c12; do timeout 200 python tools/small_proof_probe.py $k 30 2>&1 | grep "ms per proof" >> $out; done
  done
done
unset ZK_LH_COOP_BELOW
cat $out
