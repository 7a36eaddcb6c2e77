#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_msm.py -m gpu -x -q > gpurun_out/s_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s_pytest.log; tail -5 gpurun_out/s_pytest.log
for c in 1 2 4 8; do echo "chunks $c"; ZK_MSM_CHUNKS=$c timeout 300 python tools/msm_bench.py bn254 g1 22 2>/dev/null | tail -1; ZK_MSM_CHUNKS=$c timeout 300 python tools/msm_bench.py bls12_381 g1 22 2>/dev/null | tail -1; done
ZK_MSM_CHUNKS=4 timeout 300 python tools/msm_bench.py bn254 g1 20 2>/dev/null | tail -1
ZK_MSM_CHUNKS=1 timeout 300 python tools/msm_bench.py bn254 g1 20 2>/dev/null | tail -1
