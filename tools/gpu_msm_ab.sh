#!/bin/bash
# MSM: the endomorphism split chunk by chunk (default) against the whole split in front of the sum (ZK_MSM_SPLIT_WHOLE=1), one box
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/msm_ab.txt; : > $out
for r in 1 2; do
  echo "== chunked split (run $r)" >> $out; ZK_MSM_SPLIT_CHUNKED=1 timeout 300 python tools/msm_bench.py bn254 g1 20 22 23 2>&1 | grep msm >> $out
  echo "== whole split (run $r)" >> $out; timeout 300 python tools/msm_bench.py bn254 g1 20 22 23 2>&1 | grep msm >> $out
done
echo "== bls12_381 chunked / whole" >> $out
ZK_MSM_SPLIT_CHUNKED=1 timeout 300 python tools/msm_bench.py bls12_381 g1 22 2>&1 | grep msm >> $out
timeout 300 python tools/msm_bench.py bls12_381 g1 22 2>&1 | grep msm >> $out
cat $out
