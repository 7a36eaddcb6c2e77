#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/m_prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/m_prof -o p -- python3 tools/prove_bench.py --nbits 24 --reps 2 > gpurun_out/m_prof.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/m_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "evals_partial" in r["Kernel_Name"] or "lev_pow" in r["Kernel_Name"] or "evals_final" in r["Kernel_Name"]]
for r in rows[-12:]:
    print(r["Kernel_Name"][:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
find gpurun_out/m_prof -name '*kernel_trace.csv' -delete
