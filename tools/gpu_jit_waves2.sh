#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/jit_waves_poseidong.txt; : > $out
for r in 1 2; do
  for w in 1 2 3; do
    echo "== ZK_JIT_WAVES=$w (run $r)" >> $out
    ZK_JIT_WAVES=$w ZK_STARK_TIMING=quiet timeout 900 python tools/prove_bench.py --nbits 22 --reps 4 2>/dev/null | grep -o '"stark_gen_ms": \[[^]]*\]\|"calculate_exps_parallel": [0-9.]*' | tr '\n' ' ' >> $out; echo >> $out
  done
done
cat $out
