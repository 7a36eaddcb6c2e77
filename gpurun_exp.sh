#!/bin/bash
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stark_prove.py tests/test_gpu_stark_steps.py tests/test_gpu_stark_concurrent.py -x -q -m gpu 2>&1 | tail -3
for k in fib c12 r1; do python3 tools/small_proof_probe.py $k 30 2>&1 | grep "ms per proof"; done
