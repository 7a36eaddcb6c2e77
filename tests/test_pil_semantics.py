"""An independent check of the code generator's reading of the PIL.

Everything the prover computes from a PIL goes through `StarkInfo` + `Program` (csrc/starkinfo_gen.hip in the product, oracle/starkinfo.py in the
checker): two restatements of `starkinfo*.rs` by one hand.  This file checks the PIL's meaning WITHOUT either of them and without any
generated program:

  * CPU: every `polIdentities` expression of a `pil.json`, evaluated straight from the expression tree on the fixture's trace, vanishes on
    every row; plookup / permutation / connection identities hold as statements about rows (every selected tuple is in the table, the two
    multisets are equal, connected cells hold equal values) -- the fixtures satisfy the PILs they ship with, read from `pil.json` alone.
  * GPU (and, with the restated CPU prover, CPU): for a proof, every evaluation the proof carries for a committed / constant / intermediate column equals Horner's rule on the
    column's own interpolation coefficients at xi (or w xi), computed here from the raw trace and the PIL's expression trees; and
    Q(xi) * Z_H(xi) == C(xi) with C built from the PIL: the identities in order under Horner's rule in the challenge vc, then one constraint
    `expression - column` per intermediate polynomial (starkinfo_cp_prover.rs:26-69, read from the Rust for this file).  The only things taken from
    the generator's output are BOOKKEEPING -- which expression ids became intermediate columns (`im_exps_list`), which evaluation sits where
    (`ev_map`), which columns are the quotient's (`qs`) -- never a generated instruction.  A wrong translation of an expression, a challenge
    index, a `next` stride or the constraint order in either generator shows up here as a mismatch.

The challenges come from replaying the transcript (stark_gen.rs:243-545 / stark_verify.rs:28-61) with the checker's sponge (oracle/, pinned by
the reference's known answers)."""
import json
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
D = ROOT / "tests" / "golden" / "starky_data"
sys.path.insert(0, str(ROOT / "tools"))
P = 0xFFFFFFFF00000001


# ---- GF(p) / GF(p^3) = GF(p)[x]/(x^3 - x - 1) on Python ints (f3g.rs:407-449) ------------------------------------------------------------
def f3(a):
    if isinstance(a, tuple):
        return a
    return (int(a) % P, 0, 0)


def f3_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P, (a[2] + b[2]) % P)
def f3_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P, (a[2] - b[2]) % P)


def f3_mul(a, b):
    a0, a1, a2 = a; b0, b1, b2 = b
    # (a0 + a1 x + a2 x^2)(b0 + b1 x + b2 x^2) with x^3 = x + 1, x^4 = x^2 + x
    c0 = a0 * b0; c1 = a0 * b1 + a1 * b0; c2 = a0 * b2 + a1 * b1 + a2 * b0; c3 = a1 * b2 + a2 * b1; c4 = a2 * b2
    return ((c0 + c3) % P, (c1 + c3 + c4) % P, (c2 + c4) % P)


def f3_pow(a, e):
    r = (1, 0, 0)
    while e:
        if e & 1:
            r = f3_mul(r, a)
        a = f3_mul(a, a); e >>= 1
    return r


def gl_root(nbits):                                                          # MG.0[nbits] (constant.rs:54-68)
    w = pow(7, 0xFFFFFFFF, P)
    for _ in range(32 - nbits):
        w = w * w % P
    return w


def interpolate(values):
    """coefficients of the polynomial of degree < N through (w^i, values[i]): a plain inverse radix-2 transform on Python ints"""
    n = len(values); nbits = n.bit_length() - 1
    a = [int(v) for v in values]
    j = 0                                                                    # bit reversal
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit; bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    winv = pow(gl_root(nbits), P - 2, P)
    length = 2
    while length <= n:
        wl = pow(winv, n // length, P)
        for s in range(0, n, length):
            w = 1
            for k in range(length // 2):
                u, v = a[s + k], a[s + k + length // 2] * w % P
                a[s + k], a[s + k + length // 2] = (u + v) % P, (u - v) % P
                w = w * wl % P
        length <<= 1
    ninv = pow(n, P - 2, P)
    return [x * ninv % P for x in a]


def horner(coefs, x):
    """sum coefs[i] x^i for base-field coefficients and x in GF(p^3)"""
    acc = (0, 0, 0)
    for c in reversed(coefs):
        acc = f3_mul(acc, x)
        acc = ((acc[0] + c) % P, acc[1], acc[2])
    return acc


# ---- the PIL's expression trees, read directly (pil-stark's pil.json: types.rs Expression) ---------------------------------------------------
class Rows:
    """evaluates expressions on the rows of a trace: a value is a list of N ints"""
    def __init__(self, pil, cm, const, n):
        self.pil, self.n = pil, n
        self.cm = [[int(v) for v in cm[j::pil["nCommitments"]]] for j in range(pil["nCommitments"])]
        self.const = [[int(v) for v in const[j::pil["nConstants"]]] for j in range(pil["nConstants"])]
        self.publics = None
        self._exp = {}

    def public_values(self):
        if self.publics is None:
            self.publics = []
            for p in self.pil["publics"]:                                     # stark_gen.rs:225-241: a cell of a committed column, or of an expression (imP)
                if p["polType"] == "cmP":
                    self.publics.append(self.cm[p["polId"]][p["idx"]])
                else:
                    self.publics.append(self.exp(p["polId"])[p["idx"]])
        return self.publics

    def exp(self, k):
        if k not in self._exp:
            self._exp[k] = self.ev(self.pil["expressions"][k])
        return self._exp[k]

    def ev(self, e):
        n, op = self.n, e["op"]
        shift = lambda col: col[1:] + col[:1]                                # next: row i reads row i + 1 (cyclic)
        if op == "cm": v = self.cm[e["id"]]; return shift(v) if e.get("next") else v
        if op == "const": v = self.const[e["id"]]; return shift(v) if e.get("next") else v
        if op == "exp": v = self.exp(e["id"]); return shift(v) if e.get("next") else v
        if op == "number": return [int(e["value"]) % P] * n
        if op == "public": return [self.public_values()[e["id"]]] * n
        if op == "neg": return [(-a) % P for a in self.ev(e["values"][0])]
        a, b = self.ev(e["values"][0]), self.ev(e["values"][1])
        if op == "add": return [(x + y) % P for x, y in zip(a, b)]
        if op == "sub": return [(x - y) % P for x, y in zip(a, b)]
        if op == "mul": return [x * y % P for x, y in zip(a, b)]
        raise ValueError("expression op " + op)


def _load(pil_name, cm_name, const_name):
    pil = json.load(open(D / pil_name))
    cm, const = np.fromfile(D / cm_name, dtype="<u8"), np.fromfile(D / const_name, dtype="<u8")
    n = cm.size // pil["nCommitments"]
    assert n * pil["nConstants"] == const.size and n & (n - 1) == 0
    return pil, cm, const, n


def test_fib_identities_vanish_on_every_row():
    pil, cm, const, n = _load("fib.pil.json", "fib.cm", "fib.const")
    R = Rows(pil, cm, const, n)
    assert len(pil["polIdentities"]) == 5
    for pi in pil["polIdentities"]:
        assert not any(R.exp(pi["e"])), "identity at fibonacci.pil line %d does not vanish" % pi["line"]
    bad = cm.copy(); bad[2 * 77] ^= 1                                         # ... and the evaluator can tell: one flipped cell breaks an identity
    Rb = Rows(pil, bad, const, n)
    assert any(any(Rb.exp(pi["e"])) for pi in pil["polIdentities"])


def test_poseidong_identities_vanish_on_every_row():
    import poseidong as PG
    nbits = 10
    pil = PG.pil(nbits)
    R = Rows(pil, PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits), PG.consts(nbits), 1 << nbits)
    assert len(pil["polIdentities"]) > 10
    for pi in pil["polIdentities"]:
        assert not any(R.exp(pi["e"])), "identity %d of poseidong.pil does not vanish" % pi["e"]


def test_plookup_rows_are_in_the_table():
    pil, cm, const, n = _load("plookup.pil.json", "plookup.cm", "plookup.const")
    R = Rows(pil, cm, const, n)
    (pl,) = pil["plookupIdentities"]
    f, t = [R.exp(k) for k in pl["f"]], [R.exp(k) for k in pl["t"]]
    sel_f, sel_t = R.exp(pl["selF"]), R.exp(pl["selT"])
    table = {tuple(c[i] for c in t) for i in range(n) if sel_t[i]}
    assert all(s in (0, 1) for s in sel_f) and any(sel_f)
    for i in range(n):
        if sel_f[i]:
            assert tuple(c[i] for c in f) in table, "row %d looks up a tuple the table does not hold" % i


def test_permutation_rows_are_the_same_multiset():
    pil, cm, const, n = _load("pe.pil.json", "pe.cm", "pe.const")
    R = Rows(pil, cm, const, n)
    (pe,) = pil["permutationIdentities"]
    f, t = [R.exp(k) for k in pe["f"]], [R.exp(k) for k in pe["t"]]
    sel_f, sel_t = R.exp(pe["selF"]), R.exp(pe["selT"])
    left = sorted(tuple(c[i] for c in f) for i in range(n) if sel_f[i])
    right = sorted(tuple(c[i] for c in t) for i in range(n) if sel_t[i])
    assert left and left == right


def test_connection_cells_hold_equal_values():
    """connection { a, b, c } is { S1, S2, S3 }: the cell (column j, row i) is wired to the cell whose identity value k_j' w^i' the constant
    S_j holds at row i (k_0 = 1, k_j = 12275445934081160404^j: helper.rs:16-23), and wired cells hold equal values"""
    pil, cm, const, n = _load("connection.pil.json", "connection.cm", "connection.const")
    R = Rows(pil, cm, const, n)
    (cn,) = pil["connectionIdentities"]
    pols, S = [R.exp(k) for k in cn["pols"]], [R.exp(k) for k in cn["connections"]]
    w, k = gl_root(n.bit_length() - 1), 12275445934081160404
    where, ks = {}, [1]
    for _ in range(len(pols) - 1):
        ks.append(ks[-1] * k % P if len(ks) > 1 else k)
    for j in range(len(pols)):
        x = ks[j]
        for i in range(n):
            where[x] = (j, i); x = x * w % P
    moved = 0
    for j in range(len(pols)):
        for i in range(n):
            jj, ii = where[S[j][i]]                                           # every S value names a cell
            moved += (jj, ii) != (j, i)
            assert pols[j][i] == pols[jj][ii], "cell (%d, %d) is wired to (%d, %d) but holds another value" % (j, i, jj, ii)
    assert moved > 0                                                          # (the fixture really wires something)


# ---- a device proof against the PIL's own semantics -------------------------------------------------------------------------------------------
class AtXi:
    """evaluates expressions at xi / w xi from the columns' evaluations (GF(p^3)); `exp` nodes that became intermediate columns read the column"""
    def __init__(self, pil, col_eval, publics, im_cm):
        self.pil, self.col_eval, self.publics, self.im_cm = pil, col_eval, publics, im_cm

    def ev(self, e, prime=False, expand_top=None):
        op = e["op"]
        pr = prime or bool(e.get("next"))
        if op == "cm": return self.col_eval[("cm", e["id"], pr)]
        if op == "const": return self.col_eval[("const", e["id"], pr)]
        if op == "exp":
            if e["id"] in self.im_cm and e["id"] != expand_top:
                return self.col_eval[("cm", self.im_cm[e["id"]], pr)]
            return self.ev(self.pil["expressions"][e["id"]], pr)
        if op == "number": return f3(int(e["value"]))
        if op == "public": return f3(self.publics[e["id"]])
        if op == "neg": return f3_sub((0, 0, 0), self.ev(e["values"][0], prime))
        a, b = self.ev(e["values"][0], prime), self.ev(e["values"][1], prime)
        return {"add": f3_add, "sub": f3_sub, "mul": f3_mul}[op](a, b)


BOOKKEEPING = ("im_exps_list", "ev_map", "qs", "q_deg", "n_cm1", "n_cm2")      # all this file takes from a generator's output (module text)


def _device_proof(zk, pil, cm, const, ss):
    """-> (bookkeeping of the PRODUCT's generator, the device proof's zkin)"""
    import importlib
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    program_json = stark.generate_program(json.dumps(pil), json.dumps(ss))
    info = json.loads(program_json)["starkinfo"]
    ns = stark.NativeStarkSetup(const, program_json, json.dumps(ss))
    z = ns.gen(cm)
    ns.free()
    return {k: info[k] for k in BOOKKEEPING}, z


def _oracle_proof(orc, pil, cm, const, ss):
    """-> (bookkeeping of the CHECKER's generator, the restated CPU prover's zkin): the same independent check pins the oracle on CPU"""
    sys.path.insert(0, str(ROOT / "oracle"))
    import stark_prover as SP, starkinfo as SI
    info, _, _ = SI.generate(json.loads(json.dumps(pil)), ss)
    su = SP.setup(json.loads(json.dumps(pil)), const, ss, orc)
    return {k: info[k] for k in BOOKKEEPING}, SP.to_zkin(SP.stark_gen(cm, su, ss, orc))


def _check_proof_against_pil(orc, pil, cm, const, ss, info, z):
    nbits = ss["nBits"]; n = 1 << nbits
    # the challenges: the transcript replayed (stark_verify.rs:28-61)
    tr = orc.transcript()
    for p in z["publics"]:
        tr.put([int(p)])
    ch = {}
    tr.put([int(v) for v in z["root1"]]); ch["u"] = tr.get_field(); ch["defVal"] = tr.get_field()
    tr.put([int(v) for v in z["root2"]]); ch["gamma"] = tr.get_field(); ch["beta"] = tr.get_field()
    tr.put([int(v) for v in z["root3"]]); vc = tuple(int(v) for v in tr.get_field())
    tr.put([int(v) for v in z["root4"]]); xi = tuple(int(v) for v in tr.get_field())
    w = gl_root(nbits)
    wxi = f3_mul(xi, f3(w))
    evals = [tuple(int(v) for v in e) + (0,) * (3 - len(e)) if isinstance(e, list) else f3(int(e)) for e in z["evals"]]
    # the columns this file can rebuild from the trace and the PIL alone: committed (stage 1), constant, intermediate (= their expression on the rows)
    R = Rows(pil, cm, const, n)
    assert [str(v) for v in R.public_values()] == [str(v) for v in z["publics"]]
    im_list = info["im_exps_list"]
    n_cm12 = info["n_cm1"] + info["n_cm2"]
    assert info["n_cm2"] == 0, "this check covers PILs without plookup / permutation / connection arguments (no Z, h1, h2 columns)"
    im_cm = {k: n_cm12 + i for i, k in enumerate(im_list)}                    # starkinfo_cp_prover.rs:56-59: the i-th (sorted) expression gets column nCommitments + i
    rows_of = {("cm", j): R.cm[j] for j in range(pil["nCommitments"])}
    rows_of.update({("const", j): R.const[j] for j in range(pil["nConstants"])})
    rows_of.update({("cm", c): R.exp(k) for k, c in im_cm.items()})
    coefs = {}
    col_eval, seen_q, checked = {}, {}, 0
    for j, m in enumerate(info["ev_map"]):
        typ, cid, prime = m["type_"], m["id"], bool(m["prime"])
        if typ == "cm" and cid in info["qs"]:
            seen_q[cid] = evals[j]
            assert not prime
            continue
        key = (typ, cid)
        assert key in rows_of, "ev_map names a column this PIL does not have: %r" % (m,)
        if key not in coefs:
            coefs[key] = interpolate(rows_of[key])
        want = horner(coefs[key], wxi if prime else xi)
        assert evals[j] == want, "evals[%d] (%s %d%s) is not the column's value at %s" % (j, typ, cid, "'" if prime else "", "w xi" if prime else "xi")
        col_eval[(typ, cid, prime)] = want
        checked += 1
    assert checked + len(seen_q) == len(evals) and len(seen_q) == info["q_deg"]
    # C(xi): identities under Horner's rule in vc, then `expression - column` per intermediate polynomial (starkinfo_cp_prover.rs:26-69)
    A = AtXi(pil, col_eval, R.public_values(), im_cm)
    C = None
    for pi in pil["polIdentities"]:
        e = A.ev({"op": "exp", "id": pi["e"]})
        C = e if C is None else f3_add(f3_mul(vc, C), e)
    for k in im_list:
        e = f3_sub(A.ev({"op": "exp", "id": k}, expand_top=k), col_eval[("cm", im_cm[k], False)])
        C = e if C is None else f3_add(f3_mul(vc, C), e)
    x_n = f3_pow(xi, n)
    Q, acc = (0, 0, 0), (1, 0, 0)
    for q_id in info["qs"]:
        Q = f3_add(Q, f3_mul(acc, seen_q[q_id])); acc = f3_mul(acc, x_n)
    assert f3_mul(Q, f3_sub(x_n, (1, 0, 0))) == C, "Q(xi) Z_H(xi) != C(xi) with C built from the PIL's identities"
    return checked, len(im_list)


@pytest.mark.gpu
@pytest.mark.parametrize("ext_bits", [1, 2])
def test_device_proof_of_fib_matches_the_pil(zk, orc, ext_bits):
    zk.init(0)
    pil, cm, const, n = _load("fib.pil.json", "fib.cm", "fib.const")
    ss = _fib_struct(ext_bits)
    checked, n_im = _check_proof_against_pil(orc, pil, cm, const, ss, *_device_proof(zk, pil, cm, const, ss))
    assert checked >= 4


@pytest.mark.gpu
def test_device_proof_of_poseidong_matches_the_pil(zk, orc):
    """BASELINE's own PIL at 2^10 rows: 19 committed + 18 constant columns, a dozen intermediate polynomials, 91 evaluations"""
    import poseidong as PG
    zk.init(0)
    nbits = 10
    pil, cm, const, ss = PG.pil(nbits), PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits), PG.consts(nbits), PG.stark_struct(nbits)
    checked, n_im = _check_proof_against_pil(orc, pil, cm, const, ss, *_device_proof(zk, pil, cm, const, ss))
    assert checked > 80 and n_im >= 6


def _fib_struct(ext_bits):
    return {"nBits": 10, "nBitsExt": 10 + ext_bits, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 10 + ext_bits}, {"nBits": 7}, {"nBits": 3}]}


@pytest.mark.parametrize("ext_bits", [1, 2])
def test_oracle_proof_of_fib_matches_the_pil(orc, ext_bits):
    """the same check on the restated CPU prover (no GPU): an independent pin on oracle/starkinfo.py + oracle/stark_prover.py"""
    pil, cm, const, n = _load("fib.pil.json", "fib.cm", "fib.const")
    ss = _fib_struct(ext_bits)
    checked, n_im = _check_proof_against_pil(orc, pil, cm, const, ss, *_oracle_proof(orc, pil, cm, const, ss))
    assert checked >= 4


def test_oracle_proof_of_poseidong_matches_the_pil(orc):
    import poseidong as PG
    nbits = 10
    pil, cm, const, ss = PG.pil(nbits), PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits), PG.consts(nbits), PG.stark_struct(nbits)
    checked, n_im = _check_proof_against_pil(orc, pil, cm, const, ss, *_oracle_proof(orc, pil, cm, const, ss))
    assert checked > 80 and n_im >= 6


def test_the_check_notices_a_wrong_constraint_order(orc):
    """the check has teeth: the same proof against a PIL whose identities are listed in another order does not satisfy Q Z_H == C"""
    pil, cm, const, n = _load("fib.pil.json", "fib.cm", "fib.const")
    ss = _fib_struct(1)
    info, z = _oracle_proof(orc, pil, cm, const, ss)
    _check_proof_against_pil(orc, pil, cm, const, ss, info, z)
    swapped = json.loads(json.dumps(pil)); swapped["polIdentities"][0], swapped["polIdentities"][1] = swapped["polIdentities"][1], swapped["polIdentities"][0]
    with pytest.raises(AssertionError, match="Q\\(xi\\) Z_H\\(xi\\) != C\\(xi\\)"):
        _check_proof_against_pil(orc, swapped, cm, const, ss, info, z)
