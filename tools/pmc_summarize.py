"""Per-kernel averages of the FETCH_SIZE / WRITE_SIZE passes written by tools/gpu_pmc.sh."""
import csv, sys, glob, collections
root = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            acc[row["Kernel_Name"].split("(")[0][-60:]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = (sum(v) / len(v), len(v))
print(f"{'kernel':62s} {'calls':>6s} {'FETCH_SIZE/launch':>18s} {'WRITE_SIZE/launch':>18s}   (raw counter units)")
for k, d in sorted(res.items()):
    f, w = d.get("FETCH_SIZE", (0, 0)), d.get("WRITE_SIZE", (0, 0))
    print(f"{k:62s} {max(f[1], w[1]):6d} {f[0]:18.1f} {w[0]:18.1f}")
