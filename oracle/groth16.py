"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Groth16 over BN254 / BLS12-381 as the reference runs it
(groth16/src/groth16.rs:88-96 -> bellman_ce::groth16::create_random_proof, a third-party dependency that is not
under /root/reference; see oracle/groth16_impl.h for what is restated and how it is pinned).  This module holds
the host-side half of the restatement:

  * the circuit the reference synthesises from a circom R1CS + witness      algebraic/src/circom_circuit.rs:94-160
  * the file formats it reads: .r1cs (algebraic/src/r1cs_file.rs:50-118, 185-270), .wtns
    (algebraic/src/reader.rs:86-137), bellman's Parameters / VerifyingKey binary (groth16/src/api.rs:545-566;
    pinned by tests/golden/groth16/verification_key*.bin <-> .json, the fixtures of json_utils.rs:351-429)
  * bellman's generate_parameters with an EXPLICIT trapdoor (tests need the trapdoor to check proofs in the
    exponent) and create_proof with explicit r, s
  * proof.json as json_utils.rs:305-315 renders it.

Points are numpy u64 word arrays (affine, Montgomery) as everywhere in tests/oracle_lib.py; None = infinity."""
import json, struct
import numpy as np

CURVES = {
    "bn254": dict(r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
                  q=21888242871839275222246405745257275088696311157297823662689037894645226208583, nl=4, s=28, json="BN128"),
    "bls12_381": dict(r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
                      q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, nl=6, s=32, json="BLS12381"),
}


def _words(x, n):
    return np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(n)], dtype=np.uint64)


def _int(w):
    return sum(int(v) << (64 * i) for i, v in enumerate(w))


class Groth16Oracle:
    def __init__(self, lib_wrapper, curve):
        """lib_wrapper: tests/oracle_lib.load(); curve: "bn254" | "bls12_381" """
        import ctypes as C
        from oracle_lib import Curve, _u64p
        self.c = CURVES[curve]; self.curve = curve
        self.r, self.q, self.nl = self.c["r"], self.c["q"], self.c["nl"]
        L = lib_wrapper.lib
        self.g1, self.g2 = Curve(L, curve), Curve(L, curve, g2=True)
        f = lambda n: getattr(L, "orc_g16_%s_%s" % (curve, n))
        self._ntt, self._quot, self._eval, self._to_m, self._from_m = f("fr_ntt"), f("quotient"), f("r1cs_eval"), f("fr_to_mont"), f("fr_from_mont")
        self._ntt.argtypes = [_u64p, C.c_uint, C.c_int, C.c_int]; self._ntt.restype = None
        self._quot.argtypes = [_u64p, _u64p, _u64p, C.c_uint]; self._quot.restype = None
        u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
        self._eval.argtypes = [_u64p, u32p, _u64p, _u64p, C.c_uint64, _u64p]; self._eval.restype = None
        for fn in (self._to_m, self._from_m):
            fn.argtypes = [_u64p, _u64p, C.c_uint64]; fn.restype = None
        self._fq_to_mont = getattr(L, "orc_%s_fq_to_mont" % curve); self._fq_to_mont.argtypes = [_u64p, _u64p]; self._fq_to_mont.restype = None
        self.omega_root = pow(7, (self.r - 1) >> self.c["s"], self.r)

    # ---- scalar field ----------------------------------------------------------------------------------------
    def fr_array(self, ints):
        """canonical n x 4 u64"""
        return np.array([[(v >> (64 * i)) & (2**64 - 1) for i in range(4)] for v in ints], dtype=np.uint64).reshape(-1, 4)
    def fr_ints(self, arr):
        return [_int(row) for row in np.asarray(arr).reshape(-1, 4)]
    def to_mont(self, canon):
        a = np.ascontiguousarray(canon, dtype=np.uint64).reshape(-1); o = np.empty_like(a); self._to_m(a, o, a.size // 4); return o.reshape(-1, 4)
    def from_mont(self, mont):
        a = np.ascontiguousarray(mont, dtype=np.uint64).reshape(-1); o = np.empty_like(a); self._from_m(a, o, a.size // 4); return o.reshape(-1, 4)
    def ntt(self, mont, inverse=False, coset=False):
        a = np.ascontiguousarray(mont, dtype=np.uint64).reshape(-1).copy()
        n = a.size // 4; assert n & (n - 1) == 0
        self._ntt(a, n.bit_length() - 1, int(inverse), int(coset)); return a.reshape(-1, 4)
    def quotient(self, a, b, c):
        a, b, c = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1).copy() for x in (a, b, c))
        n = a.size // 4
        self._quot(a, b, c, n.bit_length() - 1); return a.reshape(-1, 4)
    def omega(self, log_n):
        return pow(self.omega_root, 1 << (self.c["s"] - log_n), self.r)

    # ---- points ----------------------------------------------------------------------------------------------
    def fq_mont_words(self, x):
        o = np.zeros(self.nl, np.uint64); self._fq_to_mont(_words(x, self.nl), o); return o
    def point_from_ints(self, coords):
        return np.concatenate([self.fq_mont_words(c) for c in coords])
    def mul(self, curve, p, k):
        """[k]p; None for infinity"""
        k %= self.r
        if p is None or k == 0: return None
        o, inf = curve.scalar_mul(p, _words(k, 4))
        return None if inf else o
    def msm(self, curve, pts, ks):
        """sum [k_i] P_i over python lists (None points / zero scalars skipped)"""
        sel = [(p, k % self.r) for p, k in zip(pts, ks) if p is not None and k % self.r]
        if not sel: return None
        o, inf = curve.msm(np.concatenate([p for p, _ in sel]), self.fr_array([k for _, k in sel]).reshape(-1), c=8 if len(sel) > 64 else 4)
        return None if inf else o

    def enc_point(self, curve, p):
        """pairing_ce uncompressed encoding: big-endian canonical coordinates, G2 as x.c1 || x.c0 || y.c1 || y.c0;
        infinity = all zero with bit 6 of byte 0 set"""
        nb = 8 * self.nl
        if p is None:
            b = bytearray(nb * (4 if curve.g2 else 2)); b[0] |= 0x40; return bytes(b)
        v = curve.affine_ints(p)
        if curve.g2: v = (v[1], v[0], v[3], v[2])
        return b"".join(x.to_bytes(nb, "big") for x in v)
    def dec_point(self, curve, b):
        nb = 8 * self.nl
        if b[0] & 0x40: return None
        v = [int.from_bytes(b[i * nb:(i + 1) * nb], "big") for i in range(4 if curve.g2 else 2)]
        if curve.g2: v = [v[1], v[0], v[3], v[2]]
        return self.point_from_ints(v)

    # ---- circuit (circom_circuit.rs:94-160 + prover.rs's input constraints) ----------------------------------
    def circuit(self, r1cs):
        """-> dict: num_inputs, num_aux, rows = list of (A, B, C) after the reference's skip rule, with the
        `input_i * 0 = 0` rows bellman appends, log_m"""
        ni = 1 + r1cs["n_pub_out"] + r1cs["n_pub_in"]
        rows = [(a, b, c) for a, b, c in r1cs["constraints"] if not ((len(a) == 0 or len(b) == 0) and len(c) == 0)]
        rows = rows + [([(i, 1)], [], []) for i in range(ni)]
        log_m = 0
        while (1 << log_m) < len(rows): log_m += 1
        return dict(num_inputs=ni, num_aux=r1cs["n_wires"] - ni, n_wires=r1cs["n_wires"], rows=rows, log_m=log_m)

    def densities(self, cir):
        """bellman's DensityTracker state after synthesis: a variable is counted when it appears in an A (aux
        only) / B linear combination at all (prover.rs eval())"""
        ni = cir["num_inputs"]
        a_aux, b_in, b_aux = set(), set(), set()
        for a, b, _ in cir["rows"]:
            for j, _c in a:
                if j >= ni: a_aux.add(j)
            for j, _c in b:
                (b_in if j < ni else b_aux).add(j)
        return sorted(a_aux), sorted(b_in), sorted(b_aux)

    # ---- generate_parameters with an explicit trapdoor --------------------------------------------------------
    def setup(self, r1cs, tau, alpha, beta, gamma, delta):
        r = self.r
        cir = self.circuit(r1cs)
        m = 1 << cir["log_m"]; w = self.omega(cir["log_m"])
        zt = (pow(tau, m, r) - 1) % r
        # Lagrange basis at tau: L_i = zt * w^i / (m (tau - w^i))
        wi, den = 1, []
        pows = []
        for i in range(m):
            pows.append(wi); den.append((m * (tau - wi)) % r); wi = wi * w % r
        pref = [1] * (m + 1)
        for i in range(m): pref[i + 1] = pref[i] * den[i] % r
        inv_all = pow(pref[m], -1, r); L = [0] * m
        for i in range(m - 1, -1, -1):
            L[i] = zt * pows[i] % r * (inv_all * pref[i] % r) % r
            inv_all = inv_all * den[i] % r
        nw = cir["n_wires"]
        at, bt, ct = [0] * nw, [0] * nw, [0] * nw
        for i, (a, b, c) in enumerate(cir["rows"]):
            for j, cf in a: at[j] = (at[j] + cf * L[i]) % r
            for j, cf in b: bt[j] = (bt[j] + cf * L[i]) % r
            for j, cf in c: ct[j] = (ct[j] + cf * L[i]) % r
        G1, G2 = self.g1.generator(), self.g2.generator()
        ginv, dinv = pow(gamma, -1, r), pow(delta, -1, r)
        ni = cir["num_inputs"]
        ext = [(beta * at[j] + alpha * bt[j] + ct[j]) % r for j in range(nw)]
        P = dict(cir=cir)
        P["vk"] = dict(alpha_g1=self.mul(self.g1, G1, alpha), beta_g1=self.mul(self.g1, G1, beta), beta_g2=self.mul(self.g2, G2, beta),
                       gamma_g2=self.mul(self.g2, G2, gamma), delta_g1=self.mul(self.g1, G1, delta), delta_g2=self.mul(self.g2, G2, delta),
                       ic=[self.mul(self.g1, G1, ext[j] * ginv) for j in range(ni)])
        P["h"] = [self.mul(self.g1, G1, pow(tau, i, r) * zt % r * dinv) for i in range(m - 1)]
        P["l"] = [self.mul(self.g1, G1, ext[j] * dinv) for j in range(ni, nw)]
        # a, b_g1, b_g2: zero points are filtered out (generator.rs), which is what the density trackers index
        P["a"] = [p for p in (self.mul(self.g1, G1, at[j]) for j in range(nw)) if p is not None]
        P["b_g1"] = [p for p in (self.mul(self.g1, G1, bt[j]) for j in range(nw)) if p is not None]
        P["b_g2"] = [p for p in (self.mul(self.g2, G2, bt[j]) for j in range(nw)) if p is not None]
        P["trapdoor"] = dict(tau=tau, alpha=alpha, beta=beta, gamma=gamma, delta=delta, at=at, bt=bt, ct=ct, zt=zt)
        return P

    # ---- create_proof with explicit r, s ----------------------------------------------------------------------
    def csr(self, rows, which):
        ptr, cols, cf = [0], [], []
        for row in rows:
            for j, c in row[which]: cols.append(j); cf.append(c % self.r)
            ptr.append(len(cols))
        return np.array(ptr, np.uint64), np.array(cols, np.uint32), (self.to_mont(self.fr_array(cf)) if cf else np.zeros((0, 4), np.uint64))

    def abc(self, cir, witness):
        """per-row evaluations padded to the domain, Montgomery (m x 4 each)"""
        m = 1 << cir["log_m"]
        wm = self.to_mont(self.fr_array(witness)).reshape(-1)
        out = []
        for which in range(3):
            ptr, cols, cf = self.csr(cir["rows"], which)
            o = np.zeros(4 * m, np.uint64)
            self._eval(ptr, cols, np.ascontiguousarray(cf).reshape(-1), wm, len(cir["rows"]), o)
            out.append(o.reshape(-1, 4))
        return out

    def prove(self, P, witness, r_, s_):
        cir = P["cir"]; ni = cir["num_inputs"]; vk = P["vk"]
        a, b, c = self.abc(cir, witness)
        h = self.fr_ints(self.from_mont(self.quotient(a, b, c)))[:(1 << cir["log_m"]) - 1]
        a_aux, b_in, b_aux = self.densities(cir)
        inputs, aux = witness[:ni], witness[ni:]
        g1, g2 = self.g1, self.g2
        h_acc = self.msm(g1, P["h"], h)
        l_acc = self.msm(g1, P["l"], aux)
        a_ans = self.msm(g1, P["a"], list(inputs) + [witness[j] for j in a_aux])
        b_sc = [witness[j] for j in b_in] + [witness[j] for j in b_aux]
        b1_ans = self.msm(g1, P["b_g1"], b_sc)
        b2_ans = self.msm(g2, P["b_g2"], b_sc)
        g_a = self.msm(g1, [vk["delta_g1"], vk["alpha_g1"], a_ans], [r_, 1, 1])
        g_b = self.msm(g2, [vk["delta_g2"], vk["beta_g2"], b2_ans], [s_, 1, 1])
        g_c = self.msm(g1, [vk["delta_g1"], vk["alpha_g1"], vk["beta_g1"], a_ans, b1_ans, h_acc, l_acc], [r_ * s_, s_, r_, s_, r_, 1, 1])
        return dict(a=g_a, b=g_b, c=g_c, h=h)

    def expected_proof(self, P, witness, r_, s_):
        """The unique valid proof for (witness, r, s), computed in the exponent from the trapdoor -- no transform,
        no multi-scalar sum: A = alpha + sum w_j u_j(tau) + r delta, B = beta + sum w_j v_j(tau) + s delta,
        C = (sum_aux w_j (beta u_j + alpha v_j + w_j) + h(tau) t(tau)) / delta + A s + B r - r s delta."""
        r = self.r; T = P["trapdoor"]; cir = P["cir"]; ni = cir["num_inputs"]
        At = sum(w * u for w, u in zip(witness, T["at"])) % r
        Bt = sum(w * u for w, u in zip(witness, T["bt"])) % r
        Ct = sum(w * u for w, u in zip(witness, T["ct"])) % r
        assert T["zt"] != 0
        ht_zt = (At * Bt - Ct) % r                      # h(tau) * t(tau)
        A = (T["alpha"] + At + r_ * T["delta"]) % r
        B = (T["beta"] + Bt + s_ * T["delta"]) % r
        aux = sum(witness[j] * (T["beta"] * T["at"][j] + T["alpha"] * T["bt"][j] + T["ct"][j]) for j in range(ni, cir["n_wires"])) % r
        Cc = ((aux + ht_zt) * pow(T["delta"], -1, r) + A * s_ + B * r_ - r_ * s_ * T["delta"]) % r
        return dict(a=self.mul(self.g1, self.g1.generator(), A), b=self.mul(self.g2, self.g2.generator(), B), c=self.mul(self.g1, self.g1.generator(), Cc))

    # ---- files -----------------------------------------------------------------------------------------------
    def params_bytes(self, P):
        vk = P["vk"]; g1, g2 = self.g1, self.g2
        out = [self.enc_point(g1, vk["alpha_g1"]), self.enc_point(g1, vk["beta_g1"]), self.enc_point(g2, vk["beta_g2"]), self.enc_point(g2, vk["gamma_g2"]),
               self.enc_point(g1, vk["delta_g1"]), self.enc_point(g2, vk["delta_g2"]), struct.pack(">I", len(vk["ic"]))] + [self.enc_point(g1, p) for p in vk["ic"]]
        for key, cv in (("h", g1), ("l", g1), ("a", g1), ("b_g1", g1), ("b_g2", g2)):
            out.append(struct.pack(">I", len(P[key]))); out += [self.enc_point(cv, p) for p in P[key]]
        return b"".join(out)

    def vk_from_bytes(self, b):
        g1, g2 = self.g1, self.g2; s1, s2 = 16 * self.nl, 32 * self.nl; o = 0
        def take(cv, n):
            nonlocal o
            p = self.dec_point(cv, b[o:o + n]); o += n; return p
        vk = dict(alpha_g1=take(g1, s1), beta_g1=take(g1, s1), beta_g2=take(g2, s2), gamma_g2=take(g2, s2), delta_g1=take(g1, s1), delta_g2=take(g2, s2))
        n = struct.unpack(">I", b[o:o + 4])[0]; o += 4
        vk["ic"] = [take(g1, s1) for _ in range(n)]
        return vk, o

    def r1cs_bytes(self, r1cs):
        fs = 32
        hdr = struct.pack("<I", fs) + self.r.to_bytes(fs, "little") + struct.pack("<IIIIQI", r1cs["n_wires"], r1cs["n_pub_out"], r1cs["n_pub_in"], r1cs["n_prv_in"], r1cs["n_wires"], len(r1cs["constraints"]))
        body = []
        for row in r1cs["constraints"]:
            for lc in row:
                body.append(struct.pack("<I", len(lc)))
                for j, c in lc: body.append(struct.pack("<I", j) + (c % self.r).to_bytes(fs, "little"))
        body = b"".join(body)
        wmap = b"".join(struct.pack("<Q", i) for i in range(r1cs["n_wires"]))
        out = b"r1cs" + struct.pack("<II", 1, 3)
        for t, sec in ((1, hdr), (2, body), (3, wmap)): out += struct.pack("<IQ", t, len(sec)) + sec
        return out

    def wtns_bytes(self, witness):
        fs = 32
        out = b"wtns" + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, 4 + fs + 4) + struct.pack("<I", fs) + self.r.to_bytes(fs, "little") + struct.pack("<I", len(witness))
        out += struct.pack("<IQ", 2, len(witness) * fs) + b"".join((w % self.r).to_bytes(fs, "little") for w in witness)
        return out

    def proof_json(self, pr):
        """json_utils.rs:305-315 serialize_proof(to_hex = false)"""
        a, b, c = self.g1.affine_ints(pr["a"]), self.g2.affine_ints(pr["b"]), self.g1.affine_ints(pr["c"])
        return json.dumps({"pi_a": {"x": str(a[0]), "y": str(a[1])}, "pi_b": {"x": [str(b[0]), str(b[1])], "y": [str(b[2]), str(b[3])]},
                           "pi_c": {"x": str(c[0]), "y": str(c[1])}, "protocol": "groth16", "curve": self.c["json"]}, separators=(",", ":"))


def synthetic_r1cs(r, n_mul, n_pub=2, n_prv=3, seed=1):
    """A satisfiable circom-shaped R1CS and its witness: wire 0 = ONE, then n_pub public outputs, n_prv private
    inputs, one new wire per product constraint (lc_a * lc_b = new wire).  Also exercises the reference's corner
    cases: an unused wire (its `l` base is the point at infinity), a wire that appears only in B rows, an empty
    0 * lc = 0 row (skipped by circom_circuit.rs:147-149) and a linear row (B = ONE)."""
    import random
    rng = random.Random(seed)
    n_fixed = 1 + n_pub + n_prv
    w = [1] + [0] * n_pub + [rng.randrange(r) for _ in range(n_prv)]
    cons = []
    def lc(pool, k):
        ws = rng.sample(pool, min(k, len(pool)))
        return sorted((j, rng.choice([1, r - 1, rng.randrange(1, r), rng.randrange(1, 1000)])) for j in ws)
    ev = lambda l: sum(c * w[j] for j, c in l) % r
    defined = [0] + list(range(1 + n_pub, n_fixed))
    only_b = None
    for i in range(n_mul):
        a = lc(defined, rng.randint(1, 3)); b = lc(defined, rng.randint(1, 3))
        if i == 5 and only_b is not None: b = sorted(b + [(only_b, 3)]) if all(j != only_b for j, _ in b) else b
        if i % 7 == 3: b = [(0, 1)]                                                  # linear row
        if i < n_pub: tgt = 1 + i
        else:
            tgt = len(w); w.append(0)
        w[tgt] = ev(a) * ev(b) % r
        cons.append((a, b, [(tgt, 1)]))
        if i == 2:
            only_b = tgt                                                            # used by row 5's B only
        else:
            defined.append(tgt)
        if i == 4: cons.append(([], lc(defined, 2), []))                            # 0 * lc = 0
    w.append(rng.randrange(r))                                                      # a wire no row mentions
    r1cs = dict(n_wires=len(w), n_pub_out=n_pub, n_pub_in=0, n_prv_in=n_prv, constraints=cons)
    return r1cs, w


def verifier_inputs(g, P, proof, witness):
    """(vk, proof, public inputs) as the integer tuples oracle/pairing_bn254.groth16_verify takes; proof: dict a, b, c of
    word arrays (Groth16Oracle.prove) or the proof.json dict a device / oracle run serialised"""
    vk = P["vk"]
    out_vk = dict(alpha_g1=g.g1.affine_ints(vk["alpha_g1"]), beta_g2=g.g2.affine_ints(vk["beta_g2"]), gamma_g2=g.g2.affine_ints(vk["gamma_g2"]),
                  delta_g2=g.g2.affine_ints(vk["delta_g2"]), ic=[g.g1.affine_ints(p) for p in vk["ic"]])
    if "pi_a" in proof:
        pr = dict(a=(int(proof["pi_a"]["x"]), int(proof["pi_a"]["y"])), c=(int(proof["pi_c"]["x"]), int(proof["pi_c"]["y"])),
                  b=(int(proof["pi_b"]["x"][0]), int(proof["pi_b"]["x"][1]), int(proof["pi_b"]["y"][0]), int(proof["pi_b"]["y"][1])))
    else:
        pr = dict(a=g.g1.affine_ints(proof["a"]), b=g.g2.affine_ints(proof["b"]), c=g.g1.affine_ints(proof["c"]))
    ni = P["cir"]["num_inputs"]
    return out_vk, pr, [int(w) for w in witness[1:ni]]


def read_r1cs(b):
    """the inverse of Groth16Oracle.r1cs_bytes for files other tools wrote (algebraic/src/r1cs_file.rs:50-118, 185-270):
    -> (prime, r1cs dict); terms sorted by wire as the reference does while reading (:83)"""
    assert b[:4] == b"r1cs" and struct.unpack("<I", b[4:8])[0] == 1
    n_sec = struct.unpack("<I", b[8:12])[0]
    o, secs = 12, {}
    for _ in range(n_sec):
        t, n = struct.unpack("<IQ", b[o:o + 12]); secs[t] = b[o + 12:o + 12 + n]; o += 12 + n
    h = secs[1]
    fs = struct.unpack("<I", h[:4])[0]
    prime = int.from_bytes(h[4:4 + fs], "little")
    n_wires, n_out, n_in, n_prv, _labels, n_cons = struct.unpack("<IIIIQI", h[4 + fs:])
    c, o, cons = secs[2], 0, []
    for _ in range(n_cons):
        row = []
        for _lc in range(3):
            nv = struct.unpack("<I", c[o:o + 4])[0]; o += 4
            lc = []
            for _k in range(nv):
                lc.append((struct.unpack("<I", c[o:o + 4])[0], int.from_bytes(c[o + 4:o + 4 + fs], "little"))); o += 4 + fs
            row.append(sorted(lc, key=lambda t: t[0]))
        cons.append(tuple(row))
    return prime, dict(n_wires=n_wires, n_pub_out=n_out, n_pub_in=n_in, n_prv_in=n_prv, constraints=cons)
