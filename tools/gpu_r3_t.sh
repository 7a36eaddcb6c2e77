#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3300 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/g_pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/g_pytest_all.log
tail -10 gpurun_out/g_pytest_all.log
bash tools/gpu_profiles.sh > gpurun_out/t_profiles.log 2>&1; tail -5 gpurun_out/t_profiles.log
for k in fib c12 r1; do
  rm -rf gpurun_out/sp_$k
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_$k -o p -- python3 tools/small_proof_probe.py $k 20 > gpurun_out/sp_$k.log 2>&1
  python3 tools/trace_gaps.py $(find gpurun_out/sp_$k -name '*kernel_trace.csv' | head -1) > gpurun_out/sp_${k}_gaps.txt 2>&1; cat gpurun_out/sp_${k}_gaps.txt | head -3
  find gpurun_out/sp_$k -name '*kernel_trace.csv' -delete
done
