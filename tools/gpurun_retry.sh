#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit 3 = nothing charged): tools/gpurun_retry.sh <timeout_s> <logfile> '<command>'
t=$1; log=$2; shift 2
for i in $(seq 1 60); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "gpurun rc=$rc (attempt $i)" >> "$log"; exit $rc; fi
  sleep 45
done
echo "gpurun: still busy after 60 attempts" >> "$log"; exit 3
