#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_stark_prove.py tests/test_gpu_stark_steps.py tests/test_gpu_stark_large.py tests/test_gpu_stark_concurrent.py tests/test_gpu_c12.py tests/test_gpu_bn128.py -m gpu -x -q > gpurun_out/l_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/l_pytest.log; tail -6 gpurun_out/l_pytest.log
ZK_STARK_TIMING=1 timeout 900 python tools/prove_bench.py --nbits 20 24 --reps 3 2> gpurun_out/l_timing.log | cut -c1-300
grep "zkgpu stark_gen" gpurun_out/l_timing.log | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln.split('] ',1)[1]); print(d['nBits'], 'evals', d['evals'], 'total', d['total_gpu_ms'])"
for k in fib c12 r1; do timeout 300 python tools/small_proof_probe.py $k 30 2>/dev/null | tail -1; done
