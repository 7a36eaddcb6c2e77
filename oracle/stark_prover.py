"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's GL-hash eSTARK prover and
verifier, small sizes only (the row evaluator is pure Python):

  setup      starky/src/stark_setup.rs:27-66
  stark_gen  starky/src/stark_gen.rs:193-557 (+ helpers :575-750), operand addressing
             interpreter.rs:286-524, FRI starky/src/fri.rs:84-184
  verify     starky/src/stark_verify.rs:20-250, fri.rs:186-297
  zkin JSON  starky/src/serializer.rs:140-264, digest.rs:84-112

Heavy primitives (NTT/LDE, Poseidon, Merkle, transcript, FRI fold...) come from liboracle.so.
"""
import json
import pathlib
import sys

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent / "tests"))
import interp          # noqa: E402
import starkinfo as SI  # noqa: E402
import oracle_lib       # noqa: E402

P = 0xFFFFFFFF00000001


def parse_pil_number(s):  # types.rs:221-233
    v = int(s, 16) if s.startswith("0x") else int(s)
    return v % P


def load_trace(path, n_cols):
    if isinstance(path, np.ndarray):
        return np.ascontiguousarray(path, dtype=np.uint64).reshape(-1)
    a = np.fromfile(path, dtype="<u8")                                   # polsarray.rs:137-217
    assert a.size % n_cols == 0
    return a


# ---- operand resolution (interpreter.rs get_ref / set_ref / eval_map) ------------------------------
def resolve(node, info, dom):
    t = node["type_"]
    if t == "tmp":
        return {"kind": "tmp", "id": node["id"]}
    if t == "const":
        return {"kind": "mem", "buf": "const_" + dom, "id": node["id"], "stride": info["n_constants"], "dim": 1, "prime": node["prime"]}
    if t in ("cm", "tmpExp"):
        pol_id = (info["cm_n"] if dom == "n" else info["cm_2ns"])[node["id"]] if t == "cm" else info["tmpexp_n"][node["id"]]
        p = info["var_pol_map"][pol_id]
        return {"kind": "mem", "buf": p["section"], "id": p["section_pos"], "stride": info["map_sectionsN"][p["section"]],
                "dim": p["dim"], "prime": node["prime"]}
    if t == "q":
        return {"kind": "mem", "buf": "q_2ns", "id": node["id"], "stride": info["q_dim"], "dim": info["q_dim"], "prime": False}
    if t == "f":
        return {"kind": "mem", "buf": "f_2ns", "id": node["id"], "stride": 3, "dim": 3, "prime": False}
    if t == "number":
        return {"kind": "number", "value": parse_pil_number(node["value"])}
    if t in ("public", "challenge", "eval"):
        return {"kind": t, "id": node["id"]}
    if t in ("x", "Zi", "xDivXSubXi", "xDivXSubWXi"):
        return {"kind": t}
    raise ValueError("Invalid reference type " + t)


def compile_segment(seg, info, dom):
    """Segment.first -> [(op, dest, src0, src1)] (interpreter.rs:187-225; only `first` runs, stark_gen.rs:761-782)"""
    out = []
    for c in seg["first"]:
        src = [resolve(s, info, dom) for s in c["src"]]
        out.append((c["op"], resolve(c["dest"], info, dom), src[0], src[1] if len(src) > 1 else None))
    return out


class Ctx:
    pass


def setup(pil_json, const_path, stark_struct, orc):
    """StarkSetup::new: LDE + Merkle of the constants, then codegen."""
    n_const = pil_json["nConstants"]
    nbits, nbits_ext = stark_struct["nBits"], stark_struct["nBitsExt"]
    const_n = load_trace(const_path, n_const)
    const_2ns = orc.lde(const_n, n_const, nbits, nbits_ext)
    const_tree = orc.merkelize(const_2ns, n_const, 1 << nbits_ext)
    info, prog, _ = SI.generate(pil_json, stark_struct)
    return {"const_n": const_n, "const_2ns": const_2ns, "const_tree": const_tree, "starkinfo": info, "program": prog}


class BN128Backend:
    """Stands in for the oracle handle when verificationHashType == "BN128": the reference's
    StarkProof<MerkleTreeBN128>::stark_gen::<TranscriptBN128> (prove.rs:47-61) differs from the GL instantiation
    only through the MerkleTree and Transcript traits; everything else is delegated to the GL oracle.
    Digests are 4 raw (Montgomery) limbs; a group-proof path is [depth][16] digests (merklehash_bn128.rs:86-106)."""

    def __init__(self, orc, field="bn128"):
        """field: "bn128" (MerkleTreeBN128 / TranscriptBN128) or "bls12381" (their BLS12-381 twins)"""
        self._orc, self._h = orc, (orc.bn128() if field == "bn128" else orc.bls12381())

    def __getattr__(self, name):
        return getattr(self._orc, name)

    def merkelize(self, rows, width, height):
        return self._h.merkelize(np.asarray(rows, np.uint64), width, height).reshape(-1)

    def merkle_proof(self, nodes, height, idx):
        return self._h.merkle_proof(np.asarray(nodes, np.uint64).reshape(-1, 4), height, idx).reshape(-1)

    def root_from_proof(self, row, path, idx):
        """verify_group_proof (merklehash_bn128.rs:130-138, 260-269): leaf = hash_element_matrix(row)"""
        leaf = self._h.hash_element_matrix(np.asarray(row, np.uint64))
        return self._h.root_from_proof(np.asarray(path, np.uint64).reshape(-1, 16, 4), leaf)

    def transcript(self):
        return _BN128Transcript(self._h)

    def digest_str(self, d):
        return str(self._h.from_mont(np.asarray(d, np.uint64)))


class _BN128Transcript:
    """TranscriptBN128::put (transcript_bn128.rs:90-101): 4 words = one digest, anything else one element per word
    (stark_gen.rs absorbs publics, evals and the last polynomial word by word, roots as digests)"""

    def __init__(self, h):
        self.t = h.transcript()

    def put(self, v):
        v = np.asarray(v, np.uint64).reshape(-1)
        if v.size == 4:
            self.t.put4(v)
        else:
            for w in v:
                self.t.put1(int(w))

    def get_field(self):
        return np.array(self.t.get_field(), np.uint64)

    def get_permutations(self, n, nbits):
        return self.t.get_permutations(n, nbits)


def group_proof(orc, nodes, elements, width, height, idx):
    row = [int(v) for v in elements[idx * width:(idx + 1) * width]]
    path = orc.merkle_proof(nodes, height, idx).reshape(-1, 4)
    return row, [[int(x) for x in lvl] for lvl in path]


class _Ints:
    """numpy section seen as Python ints (the pure-Python evaluator does its arithmetic on ints)"""
    def __init__(self, a): self.a = a
    def __getitem__(self, i): return int(self.a[i])


def _powers(base, n, first=1):
    out = np.empty(n, np.uint64); v = first % P
    for k in range(n):
        out[k] = v; v = v * base % P
    return out


def stark_gen(cm_path, su, stark_struct, orc, use_c=True, lean=False, log=None):
    """cm_path: a .cm file or the trace itself (flat uint64 array).  use_c: constraint programs run through
    oracle/interp.c (same semantics as oracle/interp.py, which use_c=False selects; compared in tests/test_oracle_interp.py).
    lean: sections over the N-row domain are dropped once stage 3 is committed (nothing after it reads them) and the
    trace is not copied -- the 2^24-row golden (tools/gen_golden_full.py) has to fit the build container; same proof.
    log: callable for progress lines (roots as they appear)."""
    log = log or (lambda *a: None)
    info, prog = su["starkinfo"], su["program"]
    nbits, nbits_ext = stark_struct["nBits"], stark_struct["nBitsExt"]
    ext = nbits_ext - nbits
    N, Next = 1 << nbits, 1 << nbits_ext
    sN = info["map_sectionsN"]
    if isinstance(cm_path, np.ndarray):
        cm1 = np.ascontiguousarray(cm_path, dtype=np.uint64).reshape(-1)
        cm1 = cm1 if lean else cm1.copy()                                   # cm1_n is never a destination
    else:
        cm1 = load_trace(cm_path, info["n_cm1"])
    bufs = {"cm1_n": cm1, "const_n": np.ascontiguousarray(su["const_n"], dtype=np.uint64),
            "const_2ns": np.ascontiguousarray(su["const_2ns"], dtype=np.uint64)}
    assert len(bufs["cm1_n"]) == N * sN["cm1_n"]
    for s in ("cm2_n", "cm3_n", "tmpexp_n"):
        bufs[s] = np.zeros(sN[s] * N, np.uint64)
    for s in ("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns"):
        bufs[s] = np.zeros(sN[s] * Next, np.uint64)
    bufs["q_2ns"] = np.zeros(info["q_dim"] * Next, np.uint64)
    bufs["f_2ns"] = np.zeros(3 * Next, np.uint64)
    w, w_ext = orc.root(nbits), orc.root(nbits_ext)
    x_n, x_2ns = _powers(w, N), _powers(w_ext, Next, 49)
    zi = orc.zh_inv(nbits, ext)
    challenge = [[0, 0, 0] for _ in range(8)]
    evals = []
    publics = []
    for i, pe in enumerate(info["publics"]):                                 # stark_gen.rs:256-270
        if pe["polType"] == "cmP":
            publics.append(int(bufs["cm1_n"][pe["idx"] * sN["cm1_n"] + pe["polId"]]))
        elif pe["polType"] == "imP":                                         # calculate_exp_at_point :558-572
            code = compile_segment(prog["publics_code"][i], info, "n")
            v = interp.run_at(code, {k: _Ints(b) for k, b in bufs.items()}, N, 1, pe["idx"], publics=publics, challenges=challenge, x=_Ints(x_n))
            assert len(v) == 1
            publics.append(v[0])
        else:
            raise ValueError("Invalid public type " + pe["polType"])
    tr = orc.transcript()
    for p in publics:
        tr.put([p])
    extra = {}

    def run(seg_name, dom):
        n, nxt = (N, 1) if dom == "n" else (Next, 1 << ext)
        code = compile_segment(prog[seg_name], info, dom)
        kw = dict(publics=publics, challenges=challenge, evals=evals, x=x_n if dom == "n" else x_2ns, zi=zi,
                  xdiv=extra.get("xDivXSubXi"), xdivw=extra.get("xDivXSubWXi"))
        if use_c:
            interp.run_c(orc.lib, code, bufs, n, nxt, **kw)
            return
        used = {o["buf"] for c in code for o in c[1:] if o is not None and o["kind"] == "mem"}
        lists = {k: [int(v) for v in bufs[k]] for k in used}
        kw = {k: ([int(t) for t in v] if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        interp.run(code, lists, n, nxt, **kw)
        for k in {c[1]["buf"] for c in code if c[1]["kind"] == "mem"}:
            bufs[k][:] = np.array(lists[k], np.uint64)

    def extend_and_merkelize(sec):                                           # stark_gen.rs:709-732
        width = sN[sec + "_n"]
        e = orc.lde(bufs[sec + "_n"], width, nbits, nbits_ext) if width else np.zeros(0, np.uint64)
        bufs[sec + "_2ns"] = e
        return {"nodes": orc.merkelize(e, width, Next), "elements": e, "width": width}

    tree1 = extend_and_merkelize("cm1")
    log("root1", [int(v) for v in tree1["nodes"][-4:]])
    tr.put(tree1["nodes"][-4:])
    challenge[0] = [int(v) for v in tr.get_field()]
    challenge[1] = [int(v) for v in tr.get_field()]
    run("step2prev", "n")
    n_cm = info["n_cm1"]
    for pu in info["pu_ctx"]:                                               # stark_gen.rs:300-308
        f = get_pol(bufs, info, info["exp2pol"][pu["f_exp_id"]], N).reshape(-1, 3)
        t = get_pol(bufs, info, info["exp2pol"][pu["t_exp_id"]], N).reshape(-1, 3)
        h1, h2 = calculate_h1h2([tuple(int(v) for v in r) for r in f], [tuple(int(v) for v in r) for r in t])
        set_pol(bufs, info, info["cm_n"][n_cm], [v for r in h1 for v in r], N); n_cm += 1
        set_pol(bufs, info, info["cm_n"][n_cm], [v for r in h2 for v in r], N); n_cm += 1
    tree2 = extend_and_merkelize("cm2")
    tr.put(tree2["nodes"][-4:])
    challenge[2] = [int(v) for v in tr.get_field()]
    challenge[3] = [int(v) for v in tr.get_field()]
    bufs["tmpexp_n"][:] = 0                             # output-only section of the step: starts from zero (stark_gen.rs:944-951)
    run("step3prev", "n")
    n_cm = info["n_cm1"] + info["n_cm2"]
    for o in info["pu_ctx"] + info["pe_ctx"] + info["ci_ctx"]:               # stark_gen.rs:329-353
        num, den = get_pol(bufs, info, info["exp2pol"][o["num_id"]], N), get_pol(bufs, info, info["exp2pol"][o["den_id"]], N)
        z, ok = orc.calculate_z(num, den)
        assert ok, "z does not close"
        set_pol(bufs, info, info["cm_n"][n_cm], z, N)
        n_cm += 1
    bufs["tmpexp_n"][:] = 0
    run("step3", "n")
    tree3 = extend_and_merkelize("cm3")
    log("root3", [int(v) for v in tree3["nodes"][-4:]])
    if lean:
        for k in ("cm1_n", "cm2_n", "cm3_n", "tmpexp_n", "const_n"):
            bufs[k] = np.zeros(0, np.uint64)
        cm1 = None
    tr.put(tree3["nodes"][-4:])
    challenge[4] = [int(v) for v in tr.get_field()]
    run("step42ns", "2ns")
    q_dim, q_deg = info["q_dim"], info["q_deg"]
    qq1 = orc.ntt(bufs["q_2ns"], q_dim, nbits_ext, inverse=True)             # :375-396
    qq2 = orc.qsplit(qq1, nbits, nbits_ext, q_dim, q_deg)
    cm4 = orc.ntt(qq2, q_dim * q_deg, nbits_ext) if q_deg > 0 else np.zeros(0, np.uint64)
    del qq1, qq2
    if lean:
        bufs["q_2ns"] = np.zeros(0, np.uint64)
    bufs["cm4_2ns"] = cm4
    tree4 = {"nodes": orc.merkelize(cm4, sN["cm4_2ns"], Next), "elements": cm4, "width": sN["cm4_2ns"]}
    log("root4", [int(v) for v in tree4["nodes"][-4:]])
    tr.put(tree4["nodes"][-4:])
    challenge[7] = [int(v) for v in tr.get_field()]                          # xi
    xi = np.array(challenge[7], np.uint64)
    LEv, LpEv = orc.lev(xi, nbits, False), orc.lev(xi, nbits, True)
    for ev in info["ev_map"]:                                                # :432-466
        if ev["type_"] == "const":
            buf, width, off, dim = bufs["const_2ns"], info["n_constants"], ev["id"], 1
        else:
            p = info["var_pol_map"][info["cm_2ns"][ev["id"]]]
            buf, width, off, dim = bufs[p["section"]], sN[p["section"]], p["section_pos"], p["dim"]
        evals.append([int(v) for v in orc.eval_dot(buf, width, off, dim, nbits, ext, LpEv if ev["prime"] else LEv)])
    for e in evals:
        tr.put(e)
    challenge[5] = [int(v) for v in tr.get_field()]
    challenge[6] = [int(v) for v in tr.get_field()]
    extra["xDivXSubXi"] = orc.xdivxsub(xi, nbits_ext)                        # :481-522
    wxi = np.array([int(v) * w % P for v in challenge[7]], np.uint64)
    extra["xDivXSubWXi"] = orc.xdivxsub(wxi, nbits_ext)
    run("step52ns", "2ns")
    fri_pol = bufs["f_2ns"]
    trees = [tree1, tree2, tree3, tree4, {"nodes": su["const_tree"], "elements": su["const_2ns"], "width": info["n_constants"]}]

    def query_pol(idx):
        return [group_proof(orc, t["nodes"], t["elements"], t["width"], Next, idx) for t in trees]
    fri_proof = fri_prove(orc, tr, fri_pol, stark_struct, query_pol)
    return {"rootC": [int(v) for v in su["const_tree"][-4:]], "root1": [int(v) for v in tree1["nodes"][-4:]],
            "root2": [int(v) for v in tree2["nodes"][-4:]], "root3": [int(v) for v in tree3["nodes"][-4:]],
            "root4": [int(v) for v in tree4["nodes"][-4:]], "fri_proof": fri_proof, "evals": evals, "publics": publics}


def calculate_h1h2(f, t):                                                    # stark_gen.rs:624-651
    idx_t = {}
    s = []
    for i, e in enumerate(t):
        idx_t[e] = i
        s.append((e, i))
    for e in f:
        if e not in idx_t:
            raise ValueError("Number not included: %r" % (e,))
        s.append((e, idx_t[e]))
    s.sort(key=lambda a: a[1])                                                # stable, like slice::sort_by
    return [s[2 * i][0] for i in range(len(f))], [s[2 * i + 1][0] for i in range(len(f))]


def get_pol(bufs, info, pol_id, n):                                           # stark_gen.rs:683-707
    p = info["var_pol_map"][pol_id]
    b, size, off = bufs[p["section"]], info["map_sectionsN"][p["section"]], p["section_pos"]
    out = np.zeros((n, 3), np.uint64)
    out[:, :p["dim"]] = np.asarray(b, np.uint64).reshape(n, size)[:, off:off + p["dim"]]
    return out.reshape(-1)


def set_pol(bufs, info, pol_id, pol3, n):                                     # stark_gen.rs:594-622
    p = info["var_pol_map"][pol_id]
    b, size, off = bufs[p["section"]], info["map_sectionsN"][p["section"]], p["section_pos"]
    b.reshape(n, size)[:, off:off + p["dim"]] = np.array(pol3, np.uint64).reshape(n, 3)[:, :p["dim"]]


def fri_prove(orc, tr, pol, stark_struct, query_pol):                         # fri.rs:84-184
    steps = [s["nBits"] for s in stark_struct["steps"]]
    pol_bits = stark_struct["nBitsExt"]
    shift_inv = pow(49, P - 2, P)
    trees, queries = [], [{"root": None, "pol_queries": []} for _ in steps]
    for si, step_bits in enumerate(steps):
        special_x = tr.get_field()
        pol = orc.fri_fold(pol, pol_bits, step_bits, special_x, shift_inv)
        if si < len(steps) - 1:
            nxt = steps[si + 1]
            n_groups, group_size = 1 << nxt, (1 << step_bits) >> nxt
            tb = orc.fri_transpose(pol, 1 << step_bits, nxt)
            nodes = orc.merkelize(tb, 3 * group_size, n_groups)
            trees.append({"nodes": nodes, "elements": tb, "width": 3 * group_size, "height": n_groups})
            queries[si + 1]["root"] = [int(v) for v in nodes[-4:]]
            tr.put(nodes[-4:])
        else:
            tr.put(pol)                                                        # every coefficient, 3 words each
        for _ in range(pol_bits - step_bits):
            shift_inv = shift_inv * shift_inv % P
        pol_bits = step_bits
    ys = [int(v) for v in tr.get_permutations(stark_struct["nQueries"], steps[0])]
    for si in range(len(steps)):
        for y in ys:
            if si == 0:
                queries[si]["pol_queries"].append(query_pol(y))
            else:
                t = trees[si - 1]
                queries[si]["pol_queries"].append([group_proof(orc, t["nodes"], t["elements"], t["width"], t["height"], y)])
        if si < len(steps) - 1:
            ys = [y % (1 << steps[si + 1]) for y in ys]
    return {"queries": queries, "last": [[int(v) for v in pol[3 * i:3 * i + 3]] for i in range(len(pol) // 3)]}


# ---- serializer.rs:140-264 ---------------------------------------------------------------------------
def _digest(d):                                                               # digest.rs:84-112
    return str(d[0]) if d[1] == 0 and d[2] == 0 and d[3] == 0 else [str(v) for v in d]


def to_zkin_bn128(proof, backend, prover_addr=""):
    """serializer.rs:146-262 for MerkleTreeBN128: digests and siblings are decimal Fr, 16 siblings per level"""
    dg = backend.digest_str
    z = {"rootC": dg(proof["rootC"])}
    for k in ("root1", "root2", "root3", "root4"):
        z[k] = dg(proof[k])
    z["evals"] = [[str(v) for v in e] for e in proof["evals"]]
    qs = proof["fri_proof"]["queries"]
    def sib(path):                                                            # path: depth*16 rows of 4 raw limbs
        return [[dg(path[16 * l + k]) for k in range(16)] for l in range(len(path) // 16)]
    for i in range(1, len(qs)):
        z["s%d_root" % i] = dg(qs[i]["root"])
        z["s%d_vals" % i] = [[str(v) for v in q[0][0]] for q in qs[i]["pol_queries"]]
        z["s%d_siblings" % i] = [sib(q[0][1]) for q in qs[i]["pol_queries"]]
    names = ["1", "2", "3", "4", "C"]
    for j, nm in enumerate(names):
        z["s0_vals" + nm] = [[str(v) for v in q[j][0]] for q in qs[0]["pol_queries"]]
    for j, nm in enumerate(names):
        z["s0_siblings" + nm] = [sib(q[j][1]) for q in qs[0]["pol_queries"]]
    z["finalPol"] = [[str(v) for v in e] for e in proof["fri_proof"]["last"]]
    z["publics"] = [str(p) for p in proof["publics"]]
    z["proverAddr"] = prover_addr
    return z


def to_zkin(proof):
    z = {"rootC": _digest(proof["rootC"])}
    for k in ("root1", "root2", "root3", "root4"):
        z[k] = _digest(proof[k])
    z["evals"] = [[str(v) for v in e] for e in proof["evals"]]
    qs = proof["fri_proof"]["queries"]
    sib = lambda path: [[str(v) for v in lvl] for lvl in path]                # from_basefield(v) = [v,0,0,0] -> one string
    for i in range(1, len(qs)):
        z["s%d_root" % i] = _digest(qs[i]["root"])
        z["s%d_vals" % i] = [[str(v) for v in q[0][0]] for q in qs[i]["pol_queries"]]
        z["s%d_siblings" % i] = [sib(q[0][1]) for q in qs[i]["pol_queries"]]
    names = ["1", "2", "3", "4", "C"]
    for j, nm in enumerate(names):
        z["s0_vals" + nm] = [[str(v) for v in q[j][0]] for q in qs[0]["pol_queries"]]
    for j, nm in enumerate(names):
        z["s0_siblings" + nm] = [sib(q[j][1]) for q in qs[0]["pol_queries"]]
    z["finalPol"] = [[str(v) for v in e] for e in proof["fri_proof"]["last"]]
    z["publics"] = [str(p) for p in proof["publics"]]
    return z


# ---- stark_verify.rs ---------------------------------------------------------------------------------
def from_zkin(z):
    """inverse of to_zkin (GL digests): the proof structure stark_verify takes, from the zkin JSON a prover wrote
    (serializer.rs:146-261) -- lets the restated verifier check proofs that only exist as zkin text"""
    dg = lambda d: [int(d), 0, 0, 0] if isinstance(d, str) else [int(v) for v in d]
    n_steps = 1 + sum(1 for k in z if k.startswith("s") and k.endswith("_root"))
    path = lambda p: [[int(v) for v in lvl] for lvl in p]
    n_q = len(z["s0_vals1"])
    queries = [{"root": None, "pol_queries": [[([int(v) for v in z["s0_vals" + nm][q]], path(z["s0_siblings" + nm][q]))
                                                for nm in ("1", "2", "3", "4", "C")] for q in range(n_q)]}]
    for i in range(1, n_steps):
        queries.append({"root": dg(z["s%d_root" % i]),
                        "pol_queries": [[([int(v) for v in z["s%d_vals" % i][q]], path(z["s%d_siblings" % i][q]))] for q in range(n_q)]})
    return {"rootC": dg(z["rootC"]), "root1": dg(z["root1"]), "root2": dg(z["root2"]), "root3": dg(z["root3"]), "root4": dg(z["root4"]),
            "evals": [[int(v) for v in e] for e in z["evals"]], "publics": [int(v) for v in z["publics"]],
            "fri_proof": {"queries": queries, "last": [[int(v) for v in e] for e in z["finalPol"]]}}


def from_zkin_bn128(z, backend):
    """inverse of to_zkin_bn128 (serializer.rs:146-262 with MerkleTreeBN128 / MerkleTreeBLS12381): decimal Fr digests back to
    the raw Montgomery limbs of ElementDigest<4, Fr>, 16 siblings per level -- so that the restated verifier can check a
    scalar-field-hashed proof that only exists as zkin text (the final STARK of an aggregation)"""
    dg = lambda d: [int(v) for v in backend._h.to_mont(int(d))]
    n_steps = 1 + sum(1 for k in z if k.startswith("s") and k.endswith("_root"))
    path = lambda p: [dg(p[l][k]) for l in range(len(p)) for k in range(16)]      # depth * 16 rows of 4 limbs
    n_q = len(z["s0_vals1"])
    queries = [{"root": None, "pol_queries": [[([int(v) for v in z["s0_vals" + nm][q]], path(z["s0_siblings" + nm][q]))
                                                for nm in ("1", "2", "3", "4", "C")] for q in range(n_q)]}]
    for i in range(1, n_steps):
        queries.append({"root": dg(z["s%d_root" % i]),
                        "pol_queries": [[([int(v) for v in z["s%d_vals" % i][q]], path(z["s%d_siblings" % i][q]))] for q in range(n_q)]})
    return {"rootC": dg(z["rootC"]), "root1": dg(z["root1"]), "root2": dg(z["root2"]), "root3": dg(z["root3"]), "root4": dg(z["root4"]),
            "evals": [[int(v) for v in e] for e in z["evals"]], "publics": [int(v) for v in z["publics"]],
            "fri_proof": {"queries": queries, "last": [[int(v) for v in e] for e in z["finalPol"]]}}


def f3(v):
    return tuple(int(x) for x in v)


def f3_pow(a, e):
    r = (1, 0, 0)
    while e:
        if e & 1:
            r = interp.f3_mul(r, a)
        a = interp.f3_mul(a, a); e >>= 1
    return r


def f3_inv(orc, a):
    return f3(orc.f3_inv(np.array(a, np.uint64)))


def execute_code(code, ctx):                                                  # stark_verify.rs:156-250
    tmp = {}
    def get(r):
        t = r["type_"]
        if t == "tmp": return tmp[r["id"]]
        if t in ("tree1", "tree2", "tree3", "tree4"):
            arr = ctx[t]; p = r["tree_pos"]
            return (arr[p],) if r["dim"] == 1 else (arr[p], arr[p + 1], arr[p + 2])
        if t == "const": return (ctx["consts"][r["id"]],)
        if t == "eval": return f3(ctx["evals"][r["id"]])
        if t == "number": return (parse_pil_number(r["value"]),)
        if t == "public": return (ctx["publics"][r["id"]],)
        if t == "challenge": return f3(ctx["challenge"][r["id"]])
        if t == "xDivXSubXi": return ctx["xDivXSubXi"]
        if t == "xDivXSubWXi": return ctx["xDivXSubWXi"]
        if t == "x": return f3(ctx["challenge"][7])
        if t == "Z": return ctx["Zp"] if r["prime"] else ctx["Z"]
        raise ValueError("Invalid reference type, get: " + t)
    for c in code:
        s = [get(x) for x in c["src"]]
        op = c["op"]
        if op == "add": r = interp.v_add(s[0], s[1])
        elif op == "sub": r = interp.v_sub(s[0], s[1])
        elif op == "mul": r = interp.v_mul(s[0], s[1])
        elif op == "muladd": r = interp.v_add(interp.v_mul(s[0], s[1]), s[2])
        elif op == "copy": r = s[0]
        else: raise ValueError(op)
        assert c["dest"]["type_"] == "tmp"
        tmp[c["dest"]["id"]] = r
    return get(code[-1]["dest"])


def v_eq(a, b):                                                                # f3g.rs:95-103 _eq
    a = tuple(a) + (0,) * (3 - len(a)); b = tuple(b) + (0,) * (3 - len(b))
    return a == b


def stark_verify(proof, const_root, info, prog, stark_struct, orc):
    nbits, nbits_ext = stark_struct["nBits"], stark_struct["nBitsExt"]
    N = 1 << nbits
    tr = orc.transcript()
    ch = [[0, 0, 0] for _ in range(8)]
    for p in proof["publics"]:
        tr.put([p])
    tr.put(proof["root1"]); ch[0] = f3(tr.get_field()); ch[1] = f3(tr.get_field())
    tr.put(proof["root2"]); ch[2] = f3(tr.get_field()); ch[3] = f3(tr.get_field())
    tr.put(proof["root3"]); ch[4] = f3(tr.get_field())
    tr.put(proof["root4"]); ch[7] = f3(tr.get_field())
    for e in proof["evals"]:
        tr.put(e)
    ch[5] = f3(tr.get_field()); ch[6] = f3(tr.get_field())
    w = orc.root(nbits)
    x_n = f3_pow(ch[7], N)
    ctx = {"evals": proof["evals"], "publics": proof["publics"], "challenge": ch,
           "Z": interp.v_sub(x_n, (1,)), "Zp": interp.v_sub(f3_pow(interp.v_mul(ch[7], (w,)), N), (1,))}
    res = execute_code(prog["verifier_code"]["first"], ctx)
    x_acc, q = (1,), (0,)
    for i in range(info["q_deg"]):
        q = interp.v_add(q, interp.v_mul(x_acc, f3(proof["evals"][info["ev_idx"]["cm"][(0, info["qs"][i])]])))
        x_acc = interp.v_mul(x_acc, x_n)
    if not v_eq(res, interp.v_mul(q, ctx["Z"])):
        return False
    w_ext = orc.root(nbits_ext)
    roots = [proof["root1"], proof["root2"], proof["root3"], proof["root4"], const_root]

    def check_query(query, idx):                                               # stark_verify.rs:80-136
        for (row, path), root in zip(query, roots):
            got = orc.root_from_proof(np.array(row, np.uint64), np.array(path, np.uint64).reshape(-1), idx)
            if [int(v) for v in got] != [int(v) for v in root]:
                raise ValueError("FRIVerifierFailed")
        x = 49 * pow(w_ext, idx, P) % P
        q = {"tree1": query[0][0], "tree2": query[1][0], "tree3": query[2][0], "tree4": query[3][0], "consts": query[4][0],
             "evals": proof["evals"], "publics": proof["publics"], "challenge": ch}
        q["xDivXSubXi"] = interp.v_mul((x,), f3_inv(orc, interp.v_sub((x,), ch[7])))
        q["xDivXSubWXi"] = interp.v_mul((x,), f3_inv(orc, interp.v_sub((x,), interp.v_mul(ch[7], (w,)))))
        return [execute_code(prog["verifier_query_code"]["first"], q)]
    return fri_verify(orc, tr, proof["fri_proof"], stark_struct, check_query)


def fri_verify(orc, tr, fp, stark_struct, check_query):                        # fri.rs:186-297
    steps = [s["nBits"] for s in stark_struct["steps"]]
    nq = stark_struct["nQueries"]
    special_x = []
    for si in range(len(steps)):
        special_x.append(f3(tr.get_field()))
        if si < len(steps) - 1:
            tr.put(fp["queries"][si + 1]["root"])
        else:
            for e in fp["last"]:
                tr.put(e)
    ys = [int(v) for v in tr.get_permutations(nq, steps[0])]
    pol_bits, shift = stark_struct["nBitsExt"], 49
    for si, step_bits in enumerate(steps):
        item = fp["queries"][si]
        for i in range(nq):
            if si == 0:
                pgroup = check_query(item["pol_queries"][i], ys[i])
            else:
                row, path = item["pol_queries"][i][0]
                got = orc.root_from_proof(np.array(row, np.uint64), np.array(path, np.uint64).reshape(-1), ys[i])
                if [int(v) for v in got] != item["root"]:
                    return False
                pgroup = [tuple(row[k:k + 3]) for k in range(0, len(row), 3)]                 # split3
            flat = np.array([v for e in pgroup for v in (tuple(e) + (0,) * (3 - len(e)))], np.uint64)
            bits = (len(pgroup) - 1).bit_length()
            coef = orc.f3_ntt(flat, bits, inverse=True).reshape(-1, 3)
            sinv = pow(shift * pow(orc.root(pol_bits), ys[i], P) % P, P - 2, P)
            pt = interp.v_mul(special_x[si], (sinv,))
            ev = f3(coef[-1])                                                                  # eval_pol
            for k in range(len(coef) - 2, -1, -1):
                ev = interp.v_add(interp.f3_mul(ev, pt), f3(coef[k]))
            if si < len(steps) - 1:
                nxt_groups = 1 << steps[si + 1]
                gi = ys[i] // nxt_groups
                row = fp["queries"][si + 1]["pol_queries"][i][0][0]
                if not v_eq(ev, tuple(row[3 * gi:3 * gi + 3])):
                    return False
            elif not v_eq(ev, tuple(fp["last"][ys[i]])):
                return False
        for _ in range(pol_bits - step_bits):
            shift = shift * shift % P
        pol_bits = step_bits
        if si < len(steps) - 1:
            ys = [y % (1 << steps[si + 1]) for y in ys]
    in_bits, max_deg_bits = stark_struct["nBitsExt"], stark_struct["nBits"]
    max_deg = 0 if pol_bits < (in_bits - max_deg_bits) else 1 << (pol_bits - (in_bits - max_deg_bits))
    last = np.array([v for e in fp["last"] for v in e], np.uint64)
    coef = orc.f3_ntt(last, pol_bits, inverse=True).reshape(-1, 3)
    return all(not any(int(v) for v in coef[i]) for i in range(max_deg + 1, len(coef)))


def prove_files(pil_path, const_path, cm_path, struct_path):
    orc = oracle_lib.load()
    pil = json.load(open(pil_path)); ss = json.load(open(struct_path))
    su = setup(pil, const_path, ss, orc)
    proof = stark_gen(cm_path, su, ss, orc)
    return su, proof, ss
