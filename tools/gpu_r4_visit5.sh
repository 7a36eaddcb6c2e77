#!/bin/bash
# round 4, visit 5: where the cooperative / one-lane switch of the tree levels should sit for the small proofs
mkdir -p gpurun_out/r4v5; O=gpurun_out/r4v5
for up in 32768 16384 8192 4096; do
  for k in r1 c12 fib; do
    echo "ZK_MERKLE_COOP_UPTO=$up $(ZK_MERKLE_COOP_UPTO=$up timeout 120 python tools/small_proof_probe.py $k 40 2>&1 | grep 'ms per proof')" | tee -a $O/coop_upto.txt
  done
done
for below in 16384 8192 32768; do
  for k in r1 c12; do
    echo "ZK_LH_COOP_BELOW=$below $(ZK_LH_COOP_BELOW=$below timeout 120 python tools/small_proof_probe.py $k 40 2>&1 | grep 'ms per proof')" | tee -a $O/coop_upto.txt
  done
done
