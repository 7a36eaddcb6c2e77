#!/bin/bash
# Where a 2^24 NTT pass spends its wave-cycles: four rocprofv3 --pmc runs (SQ counters only, kernel trace) over tools/ntt_time.py,
# for every library named on the command line (ZKGPU_LIB=eigen-zkvm_amd/_exp/libzkgpu_<name>.so; "shipped" = the built one).
export TMPDIR=/tmp
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_CYCLES"
S2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
S3="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
S4="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES"
for v in "$@"; do
  if [ "$v" != shipped ]; then export ZKGPU_LIB=$PWD/eigen-zkvm_amd/_exp/libzkgpu_$v.so; else unset ZKGPU_LIB; fi
  i=0
  for S in "$S1" "$S2" "$S3" "$S4"; do
    i=$((i+1)); d=gpurun_out/pmc_ntt_sq/$v/s$i; rm -rf $d; mkdir -p $d
    timeout 300 rocprofv3 --pmc $S --kernel-trace --output-format csv -d $d -o sq -- python3 tools/ntt_time.py 24 1 > $d/log.txt 2>&1
    tail -1 $d/log.txt
  done
done
python3 - "$@" <<'PY'
import csv, collections, re, glob, sys
for v in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_ntt_sq/{v}/s*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0]
            if "ntt_pass" not in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print(f"## {v}")
    for k, d in sorted(acc.items()):
        print(f"{k}  us={sum(dur[k])/len(dur[k])/1e3:.1f}")
        print("   " + "  ".join(f"{n[3:]}={sum(x)/len(x):.4g}" for n, x in sorted(d.items())))
PY
find gpurun_out/pmc_ntt_sq -name '*.csv' -delete; find gpurun_out/pmc_ntt_sq -name '*.db' -delete
