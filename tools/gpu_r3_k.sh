#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
ZK_STARK_TIMING=1 timeout 900 python tools/prove_bench.py --nbits 18 20 22 24 --reps 3 2> gpurun_out/k_timing.log > /dev/null
grep "zkgpu stark_gen" gpurun_out/k_timing.log | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln.split('] ',1)[1]); print(d['nBits'], 'open', d['openings_readback'], 'fri', d['fri_prove'], 'evals', d['evals'], 'total', d['total_gpu_ms'])"
