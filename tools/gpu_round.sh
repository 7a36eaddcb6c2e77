#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --steps 50 --warmup 5 > gpurun_out/bench.log 2>&1; tail -2 gpurun_out/bench.log
rm -rf gpurun_out/prof_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -o ntt -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-prove --no-msm --no-bn128 --no-groth16 > gpurun_out/prof_trace.log 2>&1
find gpurun_out/prof_trace -name '*stats*' | head; 
f=$(find gpurun_out/prof_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f"
