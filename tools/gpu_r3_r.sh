#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -x -q > gpurun_out/r_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r_pytest.log; tail -15 gpurun_out/r_pytest.log
