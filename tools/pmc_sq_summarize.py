"""Averages per launch of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_sq_summarize.py DIR [min_total_us] (tools/gpu_round.sh sq / icache)"""
import csv, collections, glob, re, sys
d0 = sys.argv[1]; min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0
fs = glob.glob(d0 + "/**/*counter_collection.csv", recursive=True)
if not fs:
    raise SystemExit("no counter_collection.csv under " + d0)
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for row in csv.DictReader(open(fs[0])):
    k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0].replace("zk::", "")[:52]
    key = (k, row["Grid_Size"])
    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
names = sorted({n for d in acc.values() for n in d})
print("# averages per launch; counters:", " ".join(names))
for key, d in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
    if sum(dur[key]) / 1e3 < min_us:
        continue
    g = lambda n: sum(d[n]) / len(d[n]) if d[n] else 0
    print(f"{key[0]:52s} grid={key[1]:>9s} n={len(dur[key]):4d} us={sum(dur[key])/len(dur[key])/1e3:9.1f} " + " ".join(f"{n}={g(n):.3e}" for n in names))
