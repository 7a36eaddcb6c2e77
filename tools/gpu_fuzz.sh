#!/bin/bash
# a fuzz campaign on one box (profiles/rNN/fuzz.txt): gpurun -- 'bash tools/gpu_fuzz.sh SEED0'
mkdir -p gpurun_out; export TMPDIR=/tmp
s=${1:-500}; out=gpurun_out/fuzz.txt; : > $out
timeout 500 python tools/fuzz_programs.py $((s + 1000)) $((s + 1150)) 2>&1 | tail -2 >> $out
for k in 1 2 3; do timeout 200 python tools/fuzz_primitives.py $((s + 30 + k)) 3000 2>&1 | tail -1 >> $out; done
for k in 1 2 3; do timeout 400 python tools/fuzz_proofs.py $((s + 10 + k)) 100 2>&1 | tail -1 >> $out; done
timeout 600 python tools/fuzz_verify.py $((s + 20)) 150 2>&1 | tail -1 >> $out
cat $out
