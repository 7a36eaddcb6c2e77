#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" nttilp nttit; do
  if [ -n "$v" ]; then export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; else unset ZKGPU_LIB; fi
  echo "== ntt variant '$v'"; timeout 300 python tools/ntt_time.py 24 1 2>/dev/null | tail -2; timeout 300 python tools/lde_time.py 24 19 2>/dev/null | tail -1
done
for v in "" posilp posit; do
  if [ -n "$v" ]; then export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; else unset ZKGPU_LIB; fi
  echo "== poseidon variant '$v'"; timeout 300 python tools/merkle_bench.py 22 19 22 36 2>/dev/null | tail -2
done
