#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/u_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/u_pytest.log; tail -3 gpurun_out/u_pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/u_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/u_smoke.log
timeout 300 python tools/msm_bench.py bn254 g1 22 > gpurun_out/u_msm.log 2>&1; echo "msm rc=$?"; tail -1 gpurun_out/u_msm.log
