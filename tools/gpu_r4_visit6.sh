#!/bin/bash
# round 4, visit 6: the three circuits of a recursion task in a loop under the profiler (busy / idle split by tools/trace_gaps.py) + the cooperative permutation's latency
mkdir -p gpurun_out/r4v6; export TMPDIR=/tmp; O=gpurun_out/r4v6
for k in fib c12 r1; do
  rm -rf $O/sp_$k
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sp_$k -o p -- python3 tools/small_proof_probe.py $k 20 > $O/sp_$k.log 2>&1
  python3 tools/trace_gaps.py $(find $O/sp_$k -name '*kernel_trace.csv' | head -1) > $O/sp_${k}_gaps.txt 2>&1; head -3 $O/sp_${k}_gaps.txt
  cp $(find $O/sp_$k -name '*kernel_stats.csv' | head -1) $O/small_proofs_${k}_kernel_stats.csv
  find $O/sp_$k -name '*kernel_trace.csv' -delete; find $O/sp_$k -name '*.db' -delete
done
timeout 100 python tools/coop_perm_time.py | tail -1 | tee $O/coop_perm.txt
for k in fib c12 r1; do timeout 120 python tools/small_proof_probe.py $k 30 timing 2>&1 | grep -v amdgpu | tail -2 | tee -a $O/stage_timing.txt; done
