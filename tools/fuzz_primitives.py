"""One-off fuzz of the primitives against the oracle at random shapes (beyond the fixed parametrisations of tests/test_gpu_parity.py):
NTT / iNTT / LDE (any blow-up), LinearHash rows, Merkle trees of any height (odd levels) with openings walked back to the root by the
device's own path kernel (zk_stark_verify's), FRI folds.  python tools/fuzz_primitives.py SEED ROUNDS  (needs a GPU)"""
import os, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "oracle")]
import numpy as np
import eigen_zkvm_amd as zk, oracle_lib
zk.init(0)
orc = oracle_lib.load()
P = zk.P
def run(seed, rounds, verbose=True):
    global n_cases
    rng = np.random.default_rng(seed)
    R = lambda n: rng.integers(0, P, size=n, dtype=np.uint64)
    bad, t0 = [], time.time()
    n_cases = 0
    def check(name, ok, *info):
        global n_cases
        n_cases += 1
        if not ok:
            bad.append((name,) + info); print("MISMATCH", name, info, flush=True)
    for r in range(rounds):
        nbits, npols = int(rng.integers(0, 17)), int(rng.integers(1, 41))
        while (1 << nbits) * npols > (1 << 21): npols = max(1, npols // 2)
        x = R((1 << nbits) * npols)
        X = zk.fft(x, npols, nbits)
        check("fft", np.array_equal(X, orc.ntt(x, npols, nbits)), nbits, npols)
        check("ifft", np.array_equal(zk.ifft(X, npols, nbits), x), nbits, npols)
        ext = nbits + int(rng.integers(0, 4))
        if (1 << ext) * npols <= (1 << 22):
            check("lde", np.array_equal(zk.interpolate(x, npols, nbits, ext), orc.lde(x, npols, nbits, ext)), nbits, ext, npols)
        # Merkle tree of any height, openings
        h, w = int(rng.integers(1, 6000)), int(rng.integers(0, 90))
        rows = R(h * w) if w else np.zeros(0, np.uint64)
        t = zk.MerkleTreeGL(); t.merkelize(rows, w, h)
        exp = orc.merkelize(rows, w, h)
        check("merkle", np.array_equal(t.nodes(), exp), h, w)
        for idx in {0, h - 1, int(rng.integers(0, h))}:
            row, path = t.get_group_proof(idx)
            got = orc.root_from_proof(np.array(row, np.uint64), np.array(path, np.uint64).reshape(-1), idx)
            # (a one-row tree: the reference's root() is nodes[1] of get_n_nodes(1) = 2, never written -- zeros; merklehash.rs:47-61, :455-457)
            check("opening", [int(v) for v in got] == [int(v) for v in t.root()] if h > 1 else [int(v) for v in t.root()] == [0, 0, 0, 0], h, w, idx)
        t.free()
        # FRI fold
        pb = int(rng.integers(2, 15)); sb = int(rng.integers(max(0, pb - 11), pb + 1))
        pol = R(3 << pb); sx = R(3); sinv = int(rng.integers(1, P, dtype=np.uint64))
        d = zk.fri_fold(zk.DevArray.from_host(pol), pb, sb, zk.DevArray.from_host(sx), sinv)
        check("fri_fold", np.array_equal(d.to_host(), orc.fri_fold(pol, pb, sb, sx, sinv)), pb, sb)
    # scalar-field trees (MerkleTreeBN128 / BLS12381: 16-ary, any height) and G1 sums at random sizes with window-boundary scalars
    zk.bn128_init(); zk.bn128_init(field="bls12381")
    hb = {"bn128": orc.bn128(), "bls12381": orc.bls12381()}
    R254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    for r in range(max(1, rounds // 4)):
        for fld in ("bn128", "bls12381"):
            h, w = int(rng.integers(1, 700)), int(rng.integers(0, 60))
            rows = R(h * w) if w else np.zeros(0, np.uint64)
            t = zk.MerkleTreeBN128(field=fld); t.merkelize(rows, w, h)
            check("merkle_" + fld, np.array_equal(t.nodes(), hb[fld].merkelize(rows, w, h)), h, w)
            t.free()
        if r % 25 == 0:                                                         # the one-lane kernels (matrix pipe since round 6): one sponge step on more than 4096 rows,
            for fld in ("bn128", "bls12381"):                                   # ragged heights, extreme words; every 100th round a level of more than 16 384 parents
                big = r % 100 == 0
                h = int(rng.integers(262145, 270000)) if big else int(rng.integers(4097, 9000))
                w = int(rng.integers(1, 5)) if big else int(rng.integers(5, 49)) if rng.integers(0, 3) else int(rng.integers(49, 200))
                rows = R(h * w)
                for k in range(int(rng.integers(0, 6))):
                    rows[int(rng.integers(0, h * w))] = [0, P - 1, 0xFFFFFFFF, 0x8080808080808080 % P, 1][int(rng.integers(0, 5))]
                if rng.integers(0, 2):
                    rows[(h - 1) * w:] = P - 1                                  # the last row is the one idle lanes shadow
                t = zk.MerkleTreeBN128(field=fld); t.merkelize(rows, w, h)
                check("merkle_tall_" + fld, np.array_equal(t.nodes(), hb[fld].merkelize(rows, w, h)), h, w)
                t.free()
        n = int(rng.integers(1, 300))
        a, d = int(rng.integers(1, 1000)), int(rng.integers(0, 1000))
        bases = orc.bn254_make_bases(n, a, d)
        raw = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); raw[:, 3] &= np.uint64((1 << 60) - 1)
        for k in range(0, n, 7):                                                # scalars on the 16-bit window boundaries
            v = [0, 1, R254 - 1, (1 << (16 * int(rng.integers(1, 15)))) - 1, 1 << (16 * int(rng.integers(1, 15)))][int(rng.integers(0, 5))]
            raw[k] = [(v >> (64 * j)) & (2**64 - 1) for j in range(4)]
        got, inf = zk.msm_g1_bn254(bases, raw.reshape(-1))
        exp, einf = orc.bn254_msm(bases, raw.reshape(-1), 8)
        ok = inf == einf and np.array_equal(got, exp)
        if not ok and os.environ.get("ZK_FUZZ_DUMP"):                       # keep a failing sum for the bench (tools/msm_fail_probe.py)
            np.savez(os.path.join(os.environ["ZK_FUZZ_DUMP"], "msm_fail_%d_%d.npz" % (seed, n)), bases=bases, scalars=raw.reshape(-1), got=got, exp=exp, inf=inf, einf=einf, a=a, d=d)
        check("msm_bn254", ok, n)
    if verbose:
        print("fuzz primitives seed %d: %d cases in %.0f s, mismatches: %s" % (seed, n_cases, time.time() - t0, bad), flush=True)
    return bad


n_cases = 0
if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]))
