#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_stark_concurrent.py tests/test_gpu_stark_prove.py tests/test_gpu_c12.py -m gpu -x -q > gpurun_out/j_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/j_pytest.log; tail -4 gpurun_out/j_pytest.log
for k in fib c12 r1; do timeout 300 python tools/small_proof_probe.py $k 30 2>/dev/null | tail -1; done
timeout 600 python tools/stress_concurrent.py 2>&1 | tail -3
