#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
s=$(date +%s.%N); timeout 1500 python bench.py > gpurun_out/w_bench.log 2> gpurun_out/w_bench.err; rc=$?; e=$(date +%s.%N); echo "bench rc=$rc wall=$(echo "$e - $s" | bc) s"; tail -c 200 gpurun_out/w_bench.log; tail -3 gpurun_out/w_bench.err
