"""Full-size proof goldens from the CPU oracle (oracle/ only -- no GPU, no product code beyond the code generator the
oracle's own is compared with elsewhere).  Run in the build container; the outputs are small fixtures under tests/golden/:

  python tools/gen_golden_full.py 20      -> tests/golden/poseidong_2p20.json   BASELINE config 3, FRI steps 21/15/11/7/4
  python tools/gen_golden_full.py 24      -> tests/golden/poseidong_2p24.json   the headline size (about an hour, ~50 GB)
  python tools/gen_golden_full.py 24 roots   only rootC and root1 (LDE + Merkle of the constants and of the trace)

Workload = what bench.py's `stark_prove` leg and tests/test_gpu_stark_large.py prove: PoseidonG PIL, every 31-row slot
hashing its own input, `PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits)`.  A golden holds the roots, evaluations, publics,
last polynomial, FRI roots, the query indices' openings digest and sha256 of the whole zkin (`zkin_digest`: compact JSON,
keys in the serializer's order, serializer.rs:146-261) -- enough to localise a mismatch without storing the 100 kB proof.
Follows stark_gen.rs:193-557 and stark_setup.rs:27-66 through oracle/stark_prover.py.
"""
import hashlib
import json
import pathlib
import resource
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle"), str(ROOT / "tools"), str(ROOT / "tests")]


def zkin_digest(z):
    """sha256 of the proof as compact JSON in the serializer's key order -- the same function the GPU tests apply to
    the product's zkin"""
    return hashlib.sha256(json.dumps(z, separators=(",", ":")).encode()).hexdigest()


def summary(z, nbits, ss, seconds):
    keys = ["rootC", "root1", "root2", "root3", "root4", "evals", "publics", "finalPol"] + sorted(k for k in z if k.endswith("_root"))
    out = {"workload": "PoseidonG (starkjs/poseidon/poseidong.pil), trace = tools/poseidong.py trace(nbits, None, FIRST_ZERO, seed=nbits)",
           "nBits": nbits, "starkStruct": ss, "generator": "tools/gen_golden_full.py (oracle/stark_prover.py stark_gen)",
           "oracle_seconds": round(seconds, 1)}
    for k in keys:
        out[k] = z[k]
    out["openings_digest"] = {k: zkin_digest(z[k]) for k in z if k.startswith("s") and ("_vals" in k or "_siblings" in k)}
    out["zkin_digest"] = zkin_digest(z)
    return out


def main():
    import numpy as np
    import oracle_lib, poseidong as PG, stark_prover as SP
    nbits = int(sys.argv[1])
    roots_only = len(sys.argv) > 2 and sys.argv[2] == "roots"
    orc = oracle_lib.load()
    ss = PG.stark_struct(nbits)
    t0 = time.time()
    const, cm = PG.consts(nbits), PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits)
    path = ROOT / "tests" / "golden" / ("poseidong_2p%d%s.json" % (nbits, "_roots" if roots_only else ""))
    if roots_only:
        ext = ss["nBitsExt"]
        out = {"nBits": nbits, "starkStruct": ss, "generator": "tools/gen_golden_full.py roots (orc_lde + orc_merkelize)"}
        for name, a, w in (("rootC", const, PG.N_CONST), ("root1", cm, PG.N_CM)):
            e = orc.lde(a, w, nbits, ext)
            out[name] = [str(int(v)) for v in orc.merkelize(e, w, 1 << ext)[-4:]]
            del e
            print(name, out[name], "%.0f s" % (time.time() - t0), flush=True)
    else:
        su = SP.setup(PG.pil(nbits), const, ss, orc)
        print("setup %.0f s, maxrss %.1f GB" % (time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
        print("rootC", [int(v) for v in su["const_tree"][-4:]], flush=True)
        z = SP.to_zkin(SP.stark_gen(cm, su, ss, orc, lean=True, log=lambda *a: print(*a, "%.0f s" % (time.time() - t0), flush=True)))
        out = summary(z, nbits, ss, time.time() - t0)
    out["maxrss_GB"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)
    path.write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", path, "in %.0f s, maxrss %.1f GB" % (time.time() - t0, out["maxrss_GB"]))


if __name__ == "__main__":
    main()
