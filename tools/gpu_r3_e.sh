#!/bin/bash
# small-proof latency: stage timers + kernel trace per circuit
mkdir -p gpurun_out; export TMPDIR=/tmp
for k in fib c12 r1; do
  ZK_STARK_TIMING=1 timeout 300 python tools/small_proof_probe.py $k 5 2> gpurun_out/e_stages_$k.log | tail -1
  grep "stark_gen" gpurun_out/e_stages_$k.log | tail -1
done
for k in fib c12 r1; do
  rm -rf gpurun_out/e_prof_$k
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e_prof_$k -o p -- python3 tools/small_proof_probe.py $k 20 > gpurun_out/e_prof_$k.log 2>&1
  f=$(find gpurun_out/e_prof_$k -name '*kernel_stats.csv' | head -1); echo "== $k"; [ -n "$f" ] && head -25 "$f" | cut -d, -f1-6
  find gpurun_out/e_prof_$k -name '*kernel_trace.csv' -size +20M -delete
done
