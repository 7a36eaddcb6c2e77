#!/bin/bash
# where a Groth16 proof of the final wrap's size goes: gpurun -- 'bash tools/gpu_groth16_prof.sh [curve] [log_rows]'
mkdir -p gpurun_out; export TMPDIR=/tmp
c=${1:-BLS12381}; n=${2:-18}
rm -rf gpurun_out/g16
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g16 -o p -- python3 tools/groth16_bench.py $c $n > gpurun_out/g16.log 2>&1
tail -4 gpurun_out/g16.log
cp $(find gpurun_out/g16 -name '*kernel_stats.csv' | head -1) gpurun_out/groth16_${c}_2p${n}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/groth16_${c}_2p${n}_kernel_stats.csv")))
for r in rows[:28]:
    n=r['Name'].split('(')[0]
    n='::'.join(n.split('::')[-3:])
    print(f"{n[:70]:70s} calls {r['Calls']:>5s} total {int(r['TotalDurationNs'])/1e3:9.1f} us avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
python3 tools/proof_timeline.py $(find gpurun_out/g16 -name '*kernel_trace.csv' | head -1) --full 2>/dev/null | tail -5
find gpurun_out/g16 -name '*kernel_trace.csv' -delete
