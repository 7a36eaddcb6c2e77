#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stark_concurrent.py -m gpu -x -q > gpurun_out/i_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/i_pytest.log; tail -12 gpurun_out/i_pytest.log
ZK_STARK_TIMING=1 timeout 300 python tools/small_proof_probe.py r1 5 2>&1 | grep -E "stark_gen|per proof" | tail -2
ZK_STARK_TIMING=1 timeout 300 python tools/small_proof_probe.py fib 5 2>&1 | grep -E "stark_gen|per proof" | tail -2
