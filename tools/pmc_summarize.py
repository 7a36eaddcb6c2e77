"""Per-kernel averages of the FETCH_SIZE / WRITE_SIZE passes written by tools/gpu_round.sh profiles, as text and as the JSON that
bench.py's `roofline.traffic` is read from (profiles/rNN/pmc_hbm_traffic.json).

Counter units are KiB.  The 1 GiB device copy of the same run calibrates them: FETCH_SIZE reports half of a wide streaming
read on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact, so HBM bytes = (fetch_factor * FETCH + WRITE) KiB
with fetch_factor = 1 GiB / FETCH_SIZE(copy).  `src_sha16` ties the numbers to the kernel sources they were measured on."""
import collections, csv, glob, hashlib, json, pathlib, sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
NTT_SOURCES = ["ntt.hip", "ntt_reg.hip.h", "gl.hip.h"]


def ntt_src_sha16():
    h = hashlib.sha256()
    for f in NTT_SOURCES:
        h.update((ROOT / "eigen-zkvm_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def main():
    root = sys.argv[1]
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != c:
                    continue
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
                grid = int(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0)
                acc[(name, grid)].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            res[k][c] = (sum(v) / len(v), len(v))
    print(f"{'kernel':62s} {'grid':>10s} {'calls':>6s} {'FETCH_SIZE/launch':>18s} {'WRITE_SIZE/launch':>18s}   (raw counter units, KiB)")
    for k, d in sorted(res.items()):
        f, w = d.get("FETCH_SIZE", (0, 0)), d.get("WRITE_SIZE", (0, 0))
        print(f"{k[0]:62s} {k[1]:10d} {max(f[1], w[1]):6d} {f[0]:18.1f} {w[0]:18.1f}")
    copies = [d for k, d in res.items() if "copyBuffer" in k[0] and d.get("WRITE_SIZE", (0,))[0] > 1e6]
    out = {"src_sha16": ntt_src_sha16(), "units": "KiB", "calibration": None, "ntt_pass_2p24": None}
    factor = 2.0
    if copies:
        cf, cw = copies[0]["FETCH_SIZE"][0], copies[0]["WRITE_SIZE"][0]
        factor = cw / cf if cf else 2.0
        out["calibration"] = {"copy_write_kib": cw, "copy_fetch_kib": cf, "fetch_factor": round(factor, 4)}
    ntt = [(k, d) for k, d in res.items() if "ntt_pass_kernel" in k[0] and k[1] == (1 << 24) // 16 and "FETCH_SIZE" in d and "WRITE_SIZE" in d]
    if ntt:
        calls = sum(d["FETCH_SIZE"][1] for _, d in ntt)
        fetch = sum(d["FETCH_SIZE"][0] * d["FETCH_SIZE"][1] for _, d in ntt) / calls
        write = sum(d["WRITE_SIZE"][0] * d["WRITE_SIZE"][1] for _, d in ntt) / sum(d["WRITE_SIZE"][1] for _, d in ntt)
        out["ntt_pass_2p24"] = {"launches": calls, "fetch_kib": round(fetch, 1), "write_kib": round(write, 1),
                                "bytes_per_launch": round((factor * fetch + write) * 1024)}
    (pathlib.Path(root) / "pmc_hbm_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
