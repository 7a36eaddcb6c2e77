#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
timeout 900 python tools/prove_bench.py --nbits 24 --reps 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('coset', d['stark_gen_ms'])"
ZK_LDE_NO_COSET=1 timeout 900 python tools/prove_bench.py --nbits 24 --reps 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old  ', d['stark_gen_ms'])"
done
ZK_STARK_NO_OVERLAP=1 timeout 900 python tools/prove_bench.py --nbits 24 --reps 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('coset, no overlap', d['stark_gen_ms'])"
ZK_STARK_NO_OVERLAP=1 ZK_LDE_NO_COSET=1 timeout 900 python tools/prove_bench.py --nbits 24 --reps 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old, no overlap  ', d['stark_gen_ms'])"
