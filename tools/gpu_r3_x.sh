#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "poseidon or merkle or transcript or linearhash" 2>&1 | tail -2
python tools/coop_perm_time.py
for k in fib c12 r1; do timeout 200 python tools/small_proof_probe.py $k 100 2>&1 | tail -1; done
for u in 8192 16384 65536; do echo "ZK_MERKLE_COOP_UPTO=$u"; for k in c12 r1; do ZK_MERKLE_COOP_UPTO=$u timeout 200 python tools/small_proof_probe.py $k 100 2>&1 | tail -1; done; done
