#!/bin/bash
# SQ instruction-mix / VALU-utilisation counters for the NTT passes and the Poseidon kernels.
set -x
mkdir -p gpurun_out/pmc_sq
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_sq/run
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d gpurun_out/pmc_sq/run -o sq -- python3 tools/pmc_ntt.py > gpurun_out/pmc_sq/log.txt 2>&1
tail -3 gpurun_out/pmc_sq/log.txt
python3 - <<'PY'
import csv, collections, re, glob
f = glob.glob("gpurun_out/pmc_sq/run/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0][:44]
    key = (k, row["Grid_Size"])
    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for key, d in acc.items():
    if not key[0].startswith("zk::"): continue
    g = lambda n: sum(d[n]) / len(d[n]) if d[n] else 0
    print(f"{key[0]:46s} grid={key[1]:>9s} us={sum(dur[key])/len(dur[key])/1e3:9.1f} VALU={g('SQ_INSTS_VALU'):.3e} SALU={g('SQ_INSTS_SALU'):.3e} "
          f"actVALU={g('SQ_ACTIVE_INST_VALU'):.3e} actANY={g('SQ_ACTIVE_INST_ANY'):.3e} waveCyc={g('SQ_WAVE_CYCLES'):.3e} busy={g('SQ_BUSY_CYCLES'):.3e} waves={g('SQ_WAVES'):.3e} gui={g('GRBM_GUI_ACTIVE'):.3e}")
PY
