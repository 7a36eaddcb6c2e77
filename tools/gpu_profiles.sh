#!/bin/bash
# One GPU-box visit that collects everything profiles/rNN/ holds (copy the summaries from gpurun_out/ afterwards):
# bench line, rocprofv3 kernel statistics of the NTT bench / a 2^24 PoseidonG proof / the MSM, HBM-traffic counters
# (FETCH_SIZE and WRITE_SIZE in separate passes) and the SQ instruction-mix counters.
set -x
export TMPDIR=/tmp
O=gpurun_out/profiles; rm -rf $O; mkdir -p $O
timeout 1500 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ntt -o ntt -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-prove --no-msm --no-bn128 --no-groth16 --no-poseidon --no-agg > $O/ntt.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prove -o prove -- python3 tools/prove_bench.py --nbits 24 --reps 3 > $O/prove.log 2>&1; tail -1 $O/prove.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/msm -o msm -- python3 tools/msm_bench.py bn254 g1 22 > $O/msm.log 2>&1; tail -2 $O/msm.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$c -o ntt -- python3 tools/pmc_ntt.py > $O/pmc_$c.log 2>&1
done
python3 tools/pmc_summarize.py $O/pmc > $O/pmc_hbm_traffic.txt; cp $O/pmc/pmc_hbm_traffic.json $O/; tail -3 $O/pmc_hbm_traffic.txt
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $O/sq -o sq -- python3 tools/pmc_ntt.py > $O/sq.log 2>&1
python3 - <<'PY' > gpurun_out/profiles/pmc_sq.txt
import csv, collections, re, glob
f = glob.glob("gpurun_out/profiles/sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0][:44]
    key = (k, row["Grid_Size"])
    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
print("# rocprofv3 --pmc SQ_* over tools/pmc_ntt.py (2^24 NTT x8, 2^20->2^21 x20 LDE + Merkle): averages per launch")
for key, d in acc.items():
    if not key[0].startswith("zk::"): continue
    g = lambda n: sum(d[n]) / len(d[n]) if d[n] else 0
    print(f"{key[0]:46s} grid={key[1]:>9s} us={sum(dur[key])/len(dur[key])/1e3:9.1f} VALU={g('SQ_INSTS_VALU'):.3e} SALU={g('SQ_INSTS_SALU'):.3e} "
          f"actVALU={g('SQ_ACTIVE_INST_VALU'):.3e} actANY={g('SQ_ACTIVE_INST_ANY'):.3e} waveCyc={g('SQ_WAVE_CYCLES'):.3e} busy={g('SQ_BUSY_CYCLES'):.3e} waves={g('SQ_WAVES'):.3e} gui={g('GRBM_GUI_ACTIVE'):.3e}")
PY
cat gpurun_out/profiles/pmc_sq.txt | head -20
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete; find $O -name '*counter_collection.csv' -delete
du -sh $O
