"""One proof's kernel timeline out of a rocprofv3 kernel trace of tools/small_proof_probe.py: python tools/proof_timeline.py <kernel_trace.csv> [--full]
-> per kernel name: launches, busy us, the gap in front of them (us) -- for ONE steady-state proof (between two gather_proofs_kernel); --full lists every launch."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: re.sub(r"^void ", "", re.sub(r"zk::\(anonymous namespace\)::", "", r["Kernel_Name"])).split("(")[0][:44]
ends = [i for i, r in enumerate(rows[:-1]) if "gather_proofs" in r["Kernel_Name"] and "gather_proofs" not in rows[i + 1]["Kernel_Name"]]
a, b = ends[len(ends) // 2] + 1, ends[len(ends) // 2 + 1] + 1           # one proof in the middle of the run (a proof ends with its last gather_proofs_kernel)
pr = rows[a:b]
t0 = int(pr[0]["Start_Timestamp"])
agg, prev_end = {}, None
for r in pr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else max(0, s - prev_end)
    k = agg.setdefault(nm(r), [0, 0, 0])
    k[0] += 1; k[1] += e - s; k[2] += gap
    if "--full" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {nm(r)}  grid={r.get('Grid_Size', '?')} wg={r.get('Workgroup_Size', '?')}")
    prev_end = max(e, prev_end or 0)
span = int(pr[-1]["End_Timestamp"]) - t0
busy = sum(v[1] for v in agg.values()); gaps = sum(v[2] for v in agg.values())
print(f"one proof: {len(pr)} launches, span {span / 1e3:.0f} us, busy {busy / 1e3:.0f} us, gaps {gaps / 1e3:.0f} us")
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"  {k:46s} x{v[0]:3d}  busy {v[1] / 1e3:8.1f} us  gaps before {v[2] / 1e3:7.1f} us")
