#!/bin/bash
# proofs still equal the oracle's after the fused ending; latency with tuning knobs
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_stark_prove.py tests/test_gpu_stark_concurrent.py tests/test_gpu_c12.py -m gpu -x -q > gpurun_out/f_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/f_pytest.log
tail -5 gpurun_out/f_pytest.log
for k in fib c12 r1; do timeout 300 python tools/small_proof_probe.py $k 30 2>/dev/null | tail -1; done
echo "== batch upto 2^20"; ZK_LH_BATCH_UPTO=1048576 timeout 300 python tools/small_proof_probe.py r1 30 2>/dev/null | tail -1
echo "== coop upto 65536"; ZK_MERKLE_COOP_UPTO=65536 timeout 300 python tools/small_proof_probe.py r1 30 2>/dev/null | tail -1
echo "== coop upto 131072"; ZK_MERKLE_COOP_UPTO=131072 timeout 300 python tools/small_proof_probe.py r1 30 2>/dev/null | tail -1
echo "== both"; ZK_LH_BATCH_UPTO=1048576 ZK_MERKLE_COOP_UPTO=65536 timeout 300 python tools/small_proof_probe.py r1 30 2>/dev/null | tail -1
echo "== c12 coop 65536"; ZK_MERKLE_COOP_UPTO=65536 timeout 300 python tools/small_proof_probe.py c12 30 2>/dev/null | tail -1
