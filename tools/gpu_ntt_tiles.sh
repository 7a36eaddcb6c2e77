#!/bin/bash
# NTT tile granularity (profiles/r05/ntt_bound.md, experiment a): the shipped 4096-element tiles against 2048 and 1024 (tools/build_variant.sh
# tileNNNN ntt.hip "-DZK_NTT_TILE=NNNN"), 2^24 x 1 forward and inverse, each variant first checked bit-exact by the parity tests.
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/ntt_tiles.txt; : > $out
for v in shipped tile2048 tile1024; do
  if [ $v = shipped ]; then unset ZKGPU_LIB; else export ZKGPU_LIB=$PWD/eigen-zkvm_amd/variants/libzkgpu_$v.so; fi
  echo "== $v" >> $out
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ntt or lde" 2>&1 | tail -1 >> $out
  for r in 1 2; do timeout 120 python tools/ntt_time.py 24 1 >> $out 2>&1; done
done
cat $out
