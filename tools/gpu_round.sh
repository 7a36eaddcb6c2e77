#!/bin/bash
# One GPU-box visit at the end of a round (gpurun --timeout 3600 -- 'bash tools/gpu_round.sh'): the whole GPU suite, the smoke
# entry, everything profiles/rNN/ holds (tools/gpu_profiles.sh: bench line, rocprofv3 kernel statistics, HBM and SQ counters) and
# the three circuits of a recursion task in a loop under the profiler (busy / idle split by tools/trace_gaps.py).
# Outputs under gpurun_out/; copy the summaries into profiles/rNN/ afterwards.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3300 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log
tail -10 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
bash tools/gpu_profiles.sh > gpurun_out/profiles.log 2>&1; tail -5 gpurun_out/profiles.log
for k in fib c12 r1; do
  rm -rf gpurun_out/sp_$k
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_$k -o p -- python3 tools/small_proof_probe.py $k 20 > gpurun_out/sp_$k.log 2>&1
  python3 tools/trace_gaps.py $(find gpurun_out/sp_$k -name '*kernel_trace.csv' | head -1) > gpurun_out/sp_${k}_gaps.txt 2>&1; head -3 gpurun_out/sp_${k}_gaps.txt
  find gpurun_out/sp_$k -name '*kernel_trace.csv' -delete
done
timeout 100 python tools/coop_perm_time.py | tail -1
