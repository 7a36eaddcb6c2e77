"""Busy / idle split of a rocprofv3 kernel trace of one small circuit proved in a loop (tools/small_proof_probe.py):
python tools/trace_gaps.py <kernel_trace.csv> [n_proofs]  -> per proof: kernels, GPU-busy ms, idle ms, idle in gaps > 20 us and where they are"""
import csv, re, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                                            # steady state: the second half of the run
n_proofs = sum(1 for r in rows if "gather_proofs" in r["Kernel_Name"]) / max(1, len({r["Kernel_Name"] for r in rows if "gather_proofs" in r["Kernel_Name"]})) / 1.0
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
nm = lambda r: re.sub(r"^void ", "", re.sub(r"zk::\(anonymous namespace\)::", "", r["Kernel_Name"])).split("(")[0][:36]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), nm(a), nm(b)) for a, b in zip(rows, rows[1:])]
big = [g for g in gaps if g[0] > 20000]
ends = sum(1 for g in big if g[1].startswith("gather_proofs"))              # one per proof: the host work between two proofs
per = max(1, ends)
print(f"proofs in the window: {per}; per proof: {len(rows) / per:.0f} kernels, span {span / per / 1e6:.2f} ms, GPU busy {busy / per / 1e6:.2f} ms, idle {(span - busy) / per / 1e6:.2f} ms")
print(f"  idle in gaps > 20 us: {sum(g[0] for g in big) / per / 1e6:.2f} ms per proof in {len(big) / per:.1f} gaps; between two proofs (after gather_proofs_kernel): "
      f"{sum(g[0] for g in big if g[1].startswith('gather_proofs')) / per / 1e6:.2f} ms; median small gap {statistics.median([g[0] for g in gaps if 0 < g[0] <= 20000]) / 1e3:.1f} us")
where = {}
for g in big:
    if not g[1].startswith("gather_proofs"): where[(g[1], g[2])] = where.get((g[1], g[2]), 0) + g[0]
for (a, b), t in sorted(where.items(), key=lambda kv: -kv[1])[:6]:
    print(f"  {t / per / 1e3:7.1f} us per proof between {a} and {b}")
