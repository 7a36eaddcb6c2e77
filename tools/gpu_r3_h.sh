#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3300 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/g_pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/g_pytest_all.log
tail -18 gpurun_out/g_pytest_all.log
for k in fib c12 r1; do timeout 300 python tools/small_proof_probe.py $k 30 2>/dev/null | tail -1; done
