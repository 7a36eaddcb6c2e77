#!/bin/bash
# HBM-traffic counters for the NTT bench (roofline.traffic): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel trace only.  A 1 GiB device-to-device torch copy in the
# same run calibrates the counter units.  Output: gpurun_out/pmc/pmc_hbm_traffic.{txt,json} -> profiles/rNN/.
set -x
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc/$c
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/$c -o ntt -- \
      python3 tools/pmc_ntt.py > gpurun_out/pmc/$c.log 2>&1
  tail -2 gpurun_out/pmc/$c.log
done
python3 tools/pmc_summarize.py gpurun_out/pmc | tee gpurun_out/pmc/pmc_hbm_traffic.txt
find gpurun_out/pmc -name '*counter_collection.csv' -delete; find gpurun_out/pmc -name '*kernel_trace.csv' -delete; find gpurun_out/pmc -name '*.db' -delete
