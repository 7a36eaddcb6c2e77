#!/bin/bash
# SQ and cache counters of the run-time compiled constraint kernels (zk_eval_kernel) inside a 2^22-row PoseidonG proof.
set -x
export TMPDIR=/tmp
O=gpurun_out/pmc_eval; rm -rf $O; mkdir -p $O
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $O/sq -o p -- python3 tools/prove_bench.py --nbits 22 --reps 2 > $O/sq.log 2>&1
timeout 500 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum \
    --kernel-trace --output-format csv -d $O/tc -o p -- python3 tools/prove_bench.py --nbits 22 --reps 2 > $O/tc.log 2>&1
timeout 500 rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum \
    --kernel-trace --output-format csv -d $O/ta -o p -- python3 tools/prove_bench.py --nbits 22 --reps 2 > $O/ta.log 2>&1
python3 - <<'PY' > gpurun_out/pmc_eval.txt
import csv, collections, re, glob
for f in sorted(glob.glob("gpurun_out/pmc_eval/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not ("zk_eval" in k or "linearhash_rows_kernel" in k or "evals_partial" in k): continue
        key = (k.split("(")[0][-30:], row["Grid_Size"], row["VGPR_Count"])
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("#", f)
    for key, d in acc.items():
        print(key, "us=%.1f" % (sum(dur[key]) / len(dur[key]) / 1e3), " ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in sorted(d.items())))
PY
cat gpurun_out/pmc_eval.txt
find $O -name '*.db' -delete; du -sh $O
