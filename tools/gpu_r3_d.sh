#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_c12.py tests/test_gpu_stark_concurrent.py -m gpu -x -q > gpurun_out/d_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/d_pytest.log
tail -15 gpurun_out/d_pytest.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --prove-nbits 20 --no-msm --no-bn128 --no-groth16 > gpurun_out/d_bench.log 2>&1; tail -1 gpurun_out/d_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d.get('aggregation',{}); print({k:a.get(k) for k in ('tasks_per_s','s','task_latency_s')}); print(a.get('join_tree'))"
