#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stark_steps.py tests/test_gpu_stark_prove.py -m gpu -x -q 2>&1 | tail -2
for k in fib c12 r1; do timeout 200 python tools/small_proof_probe.py $k 100 2>&1 | tail -1; done
