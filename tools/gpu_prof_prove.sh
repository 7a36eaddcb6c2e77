#!/bin/bash
# rocprofv3 kernel stats of a full 2^24 proof
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_prove
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_prove -o prove -- python3 tools/prove_bench.py --nbits 24 --reps 3 > gpurun_out/prof_prove.log 2>&1
tail -3 gpurun_out/prof_prove.log
f=$(find gpurun_out/prof_prove -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -30
