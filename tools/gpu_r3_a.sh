#!/bin/bash
# round 3, visit A: Poseidon restructure -- parity + throughput
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "poseidon or linearhash or merkle or hash" > gpurun_out/a_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/a_pytest.log
tail -5 gpurun_out/a_pytest.log
timeout 300 python tools/merkle_bench.py 22 19 22 36 22 6 20 12 > gpurun_out/a_merkle.log 2>&1; cat gpurun_out/a_merkle.log
