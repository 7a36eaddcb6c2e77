#!/bin/bash
# kernel timelines of the three circuits of a recursion task (gpurun -- 'bash tools/gpu_small_proofs.sh'): outputs gpurun_out/tl_<k>.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
for k in fib c12 r1; do
  rm -rf gpurun_out/sp_$k
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sp_$k -o p -- python3 tools/small_proof_probe.py $k 12 timing > gpurun_out/sp_$k.log 2>&1
  f=$(find gpurun_out/sp_$k -name '*kernel_trace.csv' | head -1)
  python3 tools/proof_timeline.py $f > gpurun_out/tl_$k.txt 2>&1
  python3 tools/proof_timeline.py $f --full > gpurun_out/tl_${k}_full.txt 2>&1
  tail -3 gpurun_out/sp_$k.log; head -30 gpurun_out/tl_$k.txt

done
