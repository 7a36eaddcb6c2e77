#!/bin/bash
# kernel statistics + one proof's timeline of the final (BLS12381-hashed) STARK of config 5: gpurun -- 'bash tools/gpu_final_stark.sh'
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/fs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fs -o p -- python3 tools/final_stark_probe.py 8 > gpurun_out/fs.log 2>&1
tail -2 gpurun_out/fs.log
cp $(find gpurun_out/fs -name '*kernel_stats.csv' | head -1) gpurun_out/final_stark_bls12381_kernel_stats.csv
head -30 gpurun_out/final_stark_bls12381_kernel_stats.csv | cut -c1-200
python3 tools/proof_timeline.py $(find gpurun_out/fs -name '*kernel_trace.csv' | head -1) --full > gpurun_out/tl_final_full.txt 2>&1
tail -40 gpurun_out/tl_final_full.txt
ZK_STARK_TIMING=1 timeout 300 python3 tools/final_stark_probe.py 2 2>&1 | tail -3
