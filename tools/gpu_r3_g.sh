#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/g_pytest_all.log 2>&1; echo "rc=$?" >> gpurun_out/g_pytest_all.log
tail -18 gpurun_out/g_pytest_all.log
timeout 900 python bench.py > gpurun_out/g_bench.log 2>&1; tail -1 gpurun_out/g_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','ms_per_step')}, d['roofline'])
for k in ('poseidon_merkle_gl','stark_prove','msm_g1_bn254','msm_g1_bls12_381'):
    v = d.get(k) or {}; print(k, {kk: v.get(kk) for kk in ('value','ms','unit','verified','setup_s','roofline') if kk in v})
a = d.get('aggregation') or {}
print({k: a.get(k) for k in ('tasks_per_s','s','task_latency_s','scaling_ceiling')}); print(a.get('task_latency_split')); print(a.get('join_tree')); print(a.get('final_wrap'))"
