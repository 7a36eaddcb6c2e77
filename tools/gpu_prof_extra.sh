#!/bin/bash
# rocprofv3 kernel statistics of the MSM and Groth16 paths -> gpurun_out/prof_extra/*.csv (copied to profiles/rNN/ by hand)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_extra
rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/msm_g1 -o msm_g1 -- python3 $R/tools/msm_bench.py bn254 g1 22 > $O/msm_g1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/msm_g2 -o msm_g2 -- python3 $R/tools/msm_bench.py bn254 g2 22 > $O/msm_g2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g16 -o g16 -- python3 $R/tools/groth16_bench.py BN128 20 > $O/g16.log 2>&1
grep -h "msm 2\|prove" $O/*.log
find $O -name '*kernel_stats.csv'
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
