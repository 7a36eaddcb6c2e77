"""Groth16 prover timing on a synthetic circom-shaped circuit: python tools/groth16_bench.py [BN128|BLS12381] [log_rows ...]
Rows i: (w[p] + c w[q]) * w[t] = w[new_i] over earlier wires (satisfied; the quotient's top coefficient is checked
to be zero on the device result).  The proving key holds arbitrary valid points ([k]G with random 64-bit k, made
on the device): timing does not depend on the key being a real setup, proofs made here do not verify."""
import importlib, struct, sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import eigen_zkvm_amd

FR = {"BN128": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
      "BLS12381": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001}
NAME = {"BN128": "bn254", "BLS12381": "bls12_381"}


def make_circuit(r, log_rows, n_pub=1, n_prv=30, seed=1):
    """-> (r1cs bytes, witness n x 4 u64 canonical)"""
    rng = np.random.default_rng(seed)
    ni = 1 + n_pub
    n_rows = (1 << log_rows) - ni                       # + the prover's input rows = the whole domain
    first_new = ni + n_prv
    n_wires = first_new + n_rows - n_pub
    i = np.arange(n_rows)
    hi = first_new + np.maximum(i - n_pub, 0)           # wires defined before row i
    pick = lambda: (ni + np.floor(rng.random(n_rows) * (hi - ni))).astype(np.int64)
    p, q, t = pick(), pick(), pick()
    tgt = np.where(i < n_pub, 1 + i, first_new + i - n_pub)
    c = rng.integers(1, 1 << 30, size=n_rows)
    w = [0] * n_wires
    w[0] = 1
    for j in range(ni, first_new): w[j] = int(rng.integers(1, 2**62)) * int(rng.integers(1, 2**62)) % r
    for ti, pi, qi, tt, ci in zip(tgt.tolist(), p.tolist(), q.tolist(), t.tolist(), c.tolist()):
        w[ti] = (w[pi] + ci * w[qi]) % r * w[tt] % r
    term = np.dtype([("wire", "<u4"), ("coef", "<u8", 4)])
    row = np.dtype([("na", "<u4"), ("a", term, 2), ("nb", "<u4"), ("b", term, 1), ("nc", "<u4"), ("c", term, 1)])
    rec = np.zeros(n_rows, row)
    rec["na"], rec["nb"], rec["nc"] = 2, 1, 1
    rec["a"]["wire"][:, 0] = p; rec["a"]["coef"][:, 0, 0] = 1
    rec["a"]["wire"][:, 1] = q; rec["a"]["coef"][:, 1, 0] = c
    rec["b"]["wire"][:, 0] = t; rec["b"]["coef"][:, 0, 0] = 1
    rec["c"]["wire"][:, 0] = tgt; rec["c"]["coef"][:, 0, 0] = 1
    hdr = struct.pack("<I", 32) + r.to_bytes(32, "little") + struct.pack("<IIIIQI", n_wires, n_pub, 0, n_prv, n_wires, n_rows)
    body = rec.tobytes()
    wmap = np.arange(n_wires, dtype="<u8").tobytes()
    out = b"r1cs" + struct.pack("<II", 1, 3)
    for ty, sec in ((1, hdr), (2, body), (3, wmap)): out += struct.pack("<IQ", ty, len(sec)) + sec
    wit = np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in w], dtype=np.uint64)
    return out, wit, ni, n_wires


def random_points_be(zk, dev, curve, n, g2, seed):
    """n valid points as pairing_ce's uncompressed big-endian bytes"""
    if n == 0: return b""
    nl = 4 if curve == "BN128" else 6
    rng = np.random.default_rng(seed)
    k = rng.integers(1, 2**64, size=n, dtype=np.uint64)
    d = zk.g1_mul_generator(zk.DevArray.from_host(k), NAME[curve], group="g2" if g2 else "g1")
    dev.fq_convert(d, curve, to_mont=False)
    a = d.to_host().reshape(n, 4 if g2 else 2, nl)
    if g2: a = a[:, [1, 0, 3, 2], :]                     # x.c1, x.c0, y.c1, y.c0
    return a[:, :, ::-1].astype(">u8").tobytes()


def make_params(zk, dev, curve, ni, n_wires, log_rows, r1cs_density):
    na, nb = r1cs_density
    m = 1 << log_rows
    P = lambda n, g2, s: random_points_be(zk, dev, curve, n, g2, s)
    out = [P(1, False, 1), P(1, False, 2), P(1, True, 3), P(1, True, 4), P(1, False, 5), P(1, True, 6), struct.pack(">I", ni), P(ni, False, 7)]
    for n, g2, s in ((m - 1, False, 8), (n_wires - ni, False, 9), (na, False, 10), (nb, False, 11), (nb, True, 12)):
        out += [struct.pack(">I", n), P(n, g2, s)]
    return b"".join(out)


def density(r1cs_bytes, ni, n_wires):
    """(|inputs| + |aux wires in any A row|, |wires in any B row|) for the fixed row shape of make_circuit"""
    term = np.dtype([("wire", "<u4"), ("coef", "<u8", 4)])
    row = np.dtype([("na", "<u4"), ("a", term, 2), ("nb", "<u4"), ("b", term, 1), ("nc", "<u4"), ("c", term, 1)])
    off = 12 + 12 + 64 + 12                              # file header, section header, 64-byte r1cs header, section header
    n_rows = struct.unpack("<I", r1cs_bytes[84:88])[0]
    rec = np.frombuffer(r1cs_bytes, dtype=row, count=n_rows, offset=off)
    a = np.unique(rec["a"]["wire"]); b = np.unique(rec["b"]["wire"])
    return ni + int((a >= ni).sum()), int(b.size)


def main():
    args = sys.argv[1:]
    curve = args.pop(0) if args and args[0] in FR else "BN128"
    zk = eigen_zkvm_amd; zk.init(0)
    dev = importlib.import_module("eigen_zkvm_amd.groth16")
    for log_rows in [int(a) for a in args] or [16, 20]:
        t = time.perf_counter()
        rb, wit, ni, n_wires = make_circuit(FR[curve], log_rows)
        t1 = time.perf_counter()
        pb = make_params(zk, dev, curve, ni, n_wires, log_rows, density(rb, ni, n_wires))
        t2 = time.perf_counter()
        S = dev.Groth16Setup(curve, rb, pb)
        t3 = time.perf_counter()
        print(f"{curve} 2^{log_rows}: circuit {t1-t:.1f} s, key {t2-t1:.1f} s ({len(pb)/2**20:.0f} MiB), setup {t3-t2:.2f} s; wires {n_wires}", flush=True)
        assert S.domain_log == log_rows
        d_w = zk.DevArray.from_host(wit.reshape(-1))
        d_h = zk.DevArray(4 * ((1 << log_rows) - 1), zero=True)
        S.prove(d_w, 5, 7, d_h=d_h)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); S.prove(d_w, 5, 7); ts.append(time.perf_counter() - t)
        th = []
        for _ in range(3):
            t = time.perf_counter(); S.prove(wit, 5, 7); th.append(time.perf_counter() - t)
        print(f"  prove: {min(ts)*1e3:.1f} ms witness resident, {min(th)*1e3:.1f} ms from host  ({(1 << log_rows)/min(ts)/1e6:.2f} M rows/s)", flush=True)
        # transforms alone
        x = zk.DevArray.from_host(wit[:1].repeat(1 << log_rows, axis=0).reshape(-1) * 0 + np.arange(4 << log_rows, dtype=np.uint64) % 1000)
        dev.fr_ntt(x, curve); zk.lib().zk_dev_sync()
        tn = []
        for _ in range(5):
            t = time.perf_counter(); dev.fr_ntt(x, curve); zk.lib().zk_dev_sync(); tn.append(time.perf_counter() - t)
        print(f"  one transform incl. layout changes: {min(tn)*1e3:.2f} ms", flush=True)
        S.free()


if __name__ == "__main__":
    main()
