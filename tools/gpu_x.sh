#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stark_prove.py -m gpu -x -q 2>&1 | tail -2
for k in fib c12 r1; do timeout 200 python tools/small_proof_probe.py $k 100 2>&1 | tail -1; done
for k in fib c12; do
  rm -rf gpurun_out/sp_$k
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_$k -o p -- python3 tools/small_proof_probe.py $k 20 > gpurun_out/sp_$k.log 2>&1
  find gpurun_out/sp_$k -name '*kernel_trace.csv' -delete
done
