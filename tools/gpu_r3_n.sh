#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stark_large.py tests/test_gpu_stark_prove.py -m gpu -x -q > gpurun_out/n_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/n_pytest.log; tail -6 gpurun_out/n_pytest.log
ZK_STARK_TIMING=1 timeout 900 python tools/prove_bench.py --nbits 24 --reps 3 2> gpurun_out/n_timing.log | cut -c1-200
grep "zkgpu stark_gen" gpurun_out/n_timing.log | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln.split('] ',1)[1]); print(d['nBits'], 'extend', d['extend'], 'q_split', d['q_split_ntt'], 'merk', d['merkelize'], 'total', d['total_gpu_ms'])"
ZK_LDE_NO_COSET=1 ZK_STARK_TIMING=1 timeout 900 python tools/prove_bench.py --nbits 24 --reps 3 2> gpurun_out/n_timing0.log | cut -c1-100
grep "zkgpu stark_gen" gpurun_out/n_timing0.log | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln.split('] ',1)[1]); print('old', d['nBits'], 'extend', d['extend'], 'q_split', d['q_split_ntt'], 'merk', d['merkelize'], 'total', d['total_gpu_ms'])"
