#!/bin/bash
# occupancy floor of the run-time compiled step kernels (ZK_JIT_WAVES = the W of __launch_bounds__(256, W); 1 = none, as before round 5)
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/jit_waves.txt; : > $out
for r in 1 2; do
  for w in 1 2 3; do
    export ZK_JIT_WAVES=$w
    echo "== ZK_JIT_WAVES=$w (run $r)" >> $out
    for k in r1 c12; do ZK_STARK_TIMING=quiet timeout 300 python tools/small_proof_probe.py $k 30 timing 2>&1 | grep -E "ms per proof|calculate_exps" | cut -c1-400 >> $out; done
    timeout 600 ZK_STARK_TIMING=quiet python tools/prove_bench.py --nbits 22 --reps 4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('poseidong 2^22:', d['stark_gen_ms'], 'setup_s', d['setup_s'], 'exps', d.get('stages_ms', {}).get('calculate_exps_parallel'))" >> $out
  done
done
unset ZK_JIT_WAVES
cat $out
