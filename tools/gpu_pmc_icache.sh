#!/bin/bash
# instruction-cache counters of the multi-scalar sums and the scalar-field hashes: gpurun -- 'bash tools/gpu_pmc_icache.sh'
mkdir -p gpurun_out/pmc_ic; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_ic/*
run() {  # name, program...
  name=$1; shift
  timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d gpurun_out/pmc_ic/$name -o ic -- "$@" > gpurun_out/pmc_ic/$name.log 2>&1
}
run msm_bls python3 tools/msm_bench.py bls12_381 g1 20
run msm_bn python3 tools/msm_bench.py bn254 g1 20
run g16 python3 tools/groth16_bench.py BLS12381 18
run fs python3 tools/final_stark_probe.py 2
python3 - <<'PY' > gpurun_out/pmc_icache.txt
import csv, collections, re, glob
print("# rocprofv3 --pmc SQC_ICACHE_* SQ_IFETCH SQ_INSTS_VALU ...: averages per launch (tools/gpu_pmc_icache.sh)")
for name in ("msm_bn", "msm_bls", "g16", "fs"):
    fs = glob.glob("gpurun_out/pmc_ic/%s/**/*counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for row in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0].replace("zk::", "")[:52]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("==", name)
    for k, d in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
        g = lambda n: sum(d[n]) / len(d[n]) if d[n] else 0
        if sum(dur[k]) < 2e5: continue
        req, hit, mis = g("SQC_ICACHE_REQ"), g("SQC_ICACHE_HITS"), g("SQC_ICACHE_MISSES")
        print(f"{k:52s} n={len(dur[k]):4d} us={sum(dur[k])/len(dur[k])/1e3:9.1f} VALU={g('SQ_INSTS_VALU'):.3e} ifetch={g('SQ_IFETCH'):.3e} icache req={req:.3e} hit={hit:.3e} miss={mis:.3e} missrate={mis/max(req,1):.3f} waveCyc={g('SQ_WAVE_CYCLES'):.3e} busy={g('SQ_BUSY_CYCLES'):.3e}")
PY
cat gpurun_out/pmc_icache.txt | cut -c1-260
find gpurun_out/pmc_ic -name '*.csv' -delete; find gpurun_out/pmc_ic -name '*.db' -delete
