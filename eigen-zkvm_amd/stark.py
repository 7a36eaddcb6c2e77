"""Host-side driver of the GL-hash eSTARK prover on the MI355X backend.

Mirrors the reference's `starky::prove::stark_prove` flow (starky/src/prove.rs:95-160):
`StarkSetup::new` (stark_setup.rs:27-66) then `StarkProof::stark_gen` (stark_gen.rs:193-557) and
`FRI::prove` (fri.rs:84-184).  Every field operation runs in HIP kernels behind the C ABI
(include/zkgpu.h): LDE, Merkle trees, the transcript sponge, constraint evaluation (run-time
compiled step programs), Z grand products, Q split, evals, x/(x-xi), FRI folding.  The only data
that returns to the host are roots, evals, the last FRI polynomial, the query indices and the
query openings -- i.e. the proof.

Inputs: `starkinfo` / `program` exactly as the reference serialises them (serde field names of
StarkInfo starkinfo.rs:46-95, Program :27-37): the reference's own `StarkInfo::new` (or any
equivalent generator) produces them; this module does not generate code.

One host-side step remains, as in the reference: `calculate_H1H2` (stark_gen.rs:624-651, a hash-map
lookup + stable sort) -- only PILs with plookups reach it.
"""
import json

import numpy as np

from . import (DevArray, MerkleTreeGL, TranscriptGL, Program, ZkError, instr, opnd, lib, _check, _np, _ptr,
               OP_ADD, OP_SUB, OP_MUL, OP_COPY, OPND_TMP, OPND_MEM, OPND_NUMBER, OPND_PUBLIC, OPND_CHALLENGE,
               OPND_EVAL, OPND_X, OPND_ZI, OPND_XDIVXSUBXI, OPND_XDIVXSUBWXI, P)
from . import fri_fold, fri_transpose, x_table, zh_inv, xdivxsub, lev, evals as evals_dev, qsplit, EvalDesc

SLOT = {"cm1_n": 0, "cm2_n": 1, "cm3_n": 2, "tmpexp_n": 3, "const_n": 4, "cm1_2ns": 5, "cm2_2ns": 6, "cm3_2ns": 7,
        "cm4_2ns": 8, "const_2ns": 9, "q_2ns": 10, "f_2ns": 11, "scratch": 12}
OPS = {"add": OP_ADD, "sub": OP_SUB, "mul": OP_MUL, "copy": OP_COPY}


def parse_pil_number(s):  # types.rs:221-233
    v = int(s, 16) if s.startswith("0x") else int(s)
    return v % P


def _resolve(node, info, dom):
    """Node -> zk_operand, as interpreter.rs get_ref / set_ref / eval_map (:286-524) resolve addresses."""
    t = node["type_"]
    if t == "tmp":
        return opnd(OPND_TMP, id=node["id"])
    if t == "const":
        return opnd(OPND_MEM, id=node["id"], dim=1, prime=node["prime"], buf=SLOT["const_" + dom], stride=info["n_constants"])
    if t in ("cm", "tmpExp"):
        pol_id = (info["cm_n"] if dom == "n" else info["cm_2ns"])[node["id"]] if t == "cm" else info["tmpexp_n"][node["id"]]
        p = info["var_pol_map"][pol_id]
        return opnd(OPND_MEM, id=p["section_pos"], dim=p["dim"], prime=node["prime"], buf=SLOT[p["section"]],
                    stride=info["map_sectionsN"][p["section"]])
    if t == "q":
        return opnd(OPND_MEM, id=node["id"], dim=info["q_dim"], buf=SLOT["q_2ns"], stride=info["q_dim"])
    if t == "f":
        return opnd(OPND_MEM, id=node["id"], dim=3, buf=SLOT["f_2ns"], stride=3)
    if t == "number":
        return opnd(OPND_NUMBER, value=parse_pil_number(node["value"]))
    if t in ("public", "challenge", "eval"):
        return opnd({"public": OPND_PUBLIC, "challenge": OPND_CHALLENGE, "eval": OPND_EVAL}[t], id=node["id"])
    kinds = {"x": OPND_X, "Zi": OPND_ZI, "xDivXSubXi": OPND_XDIVXSUBXI, "xDivXSubWXi": OPND_XDIVXSUBWXI}
    if t in kinds:
        return opnd(kinds[t])
    raise ZkError("Invalid reference type " + t)


def compile_segment(seg, info, dom, ret_to_scratch=False):
    """compile_code (interpreter.rs:187-225): Segment.first -> one run-time compiled kernel."""
    code = []
    for c in seg["first"]:
        src = [_resolve(s, info, dom) for s in c["src"]]
        if c["op"] not in OPS:
            raise ZkError("Invalid op " + c["op"])        # the prover rejects muladd (interpreter.rs:208-216)
        code.append(instr(OPS[c["op"]], _resolve(c["dest"], info, dom), src[0], src[1] if len(src) > 1 else None))
    if ret_to_scratch and seg["first"]:                    # ret = true: the last destination is the result
        code.append(instr(OP_COPY, opnd(OPND_MEM, id=0, dim=3, buf=SLOT["scratch"], stride=3),
                          _resolve(seg["first"][-1]["dest"], info, dom)))
    return Program(code) if code else None


class StarkSetup:
    """StarkSetup::new (stark_setup.rs:27-66): LDE + Merkle tree of the constant polynomials on the
    device, and the step programs compiled for gfx950."""

    def __init__(self, const_n, starkinfo, program, stark_struct):
        self.info, self.prog, self.ss = starkinfo, program, stark_struct
        self.nbits, self.nbits_ext = stark_struct["nBits"], stark_struct["nBitsExt"]
        if stark_struct["verificationHashType"] != "GL":
            raise ZkError("only the GL hash is accelerated (SURVEY.md 8f-1)")
        nc = starkinfo["n_constants"]
        const_n = _np(const_n)
        if const_n.size != nc << self.nbits:
            raise ZkError("const trace size mismatch")
        self.d_const_n = DevArray.from_host(const_n)
        self.d_const_2ns = DevArray((nc << self.nbits_ext))
        tmp = DevArray((nc << self.nbits_ext))
        _check(lib().zk_gl_lde_dev(self.d_const_n.ptr, nc, self.nbits, self.d_const_2ns.ptr, tmp.ptr, self.nbits_ext, None))
        self.const_tree = MerkleTreeGL()
        self.const_tree.merkelize_dev(self.d_const_2ns.ptr, nc, 1 << self.nbits_ext)
        self.const_root = self.const_tree.root()
        info = starkinfo
        self.programs = {name: compile_segment(program[name], info, dom)
                         for name, dom in (("step2prev", "n"), ("step3prev", "n"), ("step3", "n"),
                                           ("step42ns", "2ns"), ("step52ns", "2ns"))}
        self.public_programs = [compile_segment(s, info, "n", ret_to_scratch=True) for s in program["publics_code"]]


def _tree_root_dev(tree):
    """device address of the root digest (last node, merklehash.rs:455-457)"""
    n_nodes = lib().zk_merkle_n_nodes(tree.height)
    return lib().zk_merkle_nodes_dev(tree._h) + 32 * (n_nodes - 1)


class _Raw:
    """a raw device pointer with a word count, duck-typed like DevArray for the binding helpers"""
    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n


def stark_gen(cm_n, setup):
    """StarkProof::stark_gen (stark_gen.rs:193-557).  Returns the proof as a dict:
    rootC, root1..4, evals, publics, fri_proof{queries[{root, pol_queries}], last}."""
    L = lib()
    info, prog, ss = setup.info, setup.prog, setup.ss
    nbits, nbits_ext = setup.nbits, setup.nbits_ext
    ext = nbits_ext - nbits
    N, Next = 1 << nbits, 1 << nbits_ext
    sN = info["map_sectionsN"]
    cm_n = _np(cm_n)
    if cm_n.size != N * sN["cm1_n"]:
        raise ZkError("cm trace size mismatch")
    B = {"cm1_n": DevArray.from_host(cm_n), "const_n": setup.d_const_n, "const_2ns": setup.d_const_2ns,
         "scratch": DevArray(3 * Next, zero=True)}
    for s in ("cm2_n", "cm3_n", "tmpexp_n"):
        B[s] = DevArray(sN[s] * N, zero=True)
    for s in ("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns"):
        B[s] = DevArray(sN[s] * Next, zero=True)
    B["q_2ns"] = DevArray(info["q_dim"] * Next, zero=True)
    B["f_2ns"] = DevArray(3 * Next, zero=True)
    x_n, x_2ns, zi = x_table(nbits, 1), x_table(nbits_ext, 49), zh_inv(nbits, ext)      # stark_gen.rs:231-249
    d_chal = DevArray(24, zero=True)                                                      # challenge[8] (constant.rs:39-50)
    n_ev = len(info["ev_map"])
    d_evals = DevArray(3 * max(1, n_ev), zero=True)
    d_pub = DevArray(max(1, len(info["publics"])), zero=True)
    dev = {"xdiv": None, "xdivw": None}

    def run(p, dom):
        if p is None:
            return
        p.run({SLOT[k]: v for k, v in B.items()}, nbits if dom == "n" else nbits_ext, 1 if dom == "n" else 1 << ext,
              publics=d_pub, challenges=d_chal, evals=d_evals, x=x_n if dom == "n" else x_2ns, zi=zi,
              xdiv=dev["xdiv"], xdivw=dev["xdivw"])

    def get_pol(pol_id):                                                                  # stark_gen.rs:683-707
        p = info["var_pol_map"][pol_id]
        out = DevArray(3 * N)
        _check(L.zk_stark_get_pol_dev(B[p["section"]].ptr, sN[p["section"]], p["section_pos"], p["dim"], N, out.ptr, None))
        return out

    def set_pol(pol_id, d_pol3):                                                          # stark_gen.rs:594-622
        p = info["var_pol_map"][pol_id]
        _check(L.zk_stark_set_pol_dev(B[p["section"]].ptr, sN[p["section"]], p["section_pos"], p["dim"], N, d_pol3.ptr, None))

    # publics (stark_gen.rs:256-270) and their absorption (:272-277)
    publics = []
    cm1_host = cm_n
    for i, pe in enumerate(info["publics"]):
        if pe["polType"] == "cmP":
            publics.append(int(cm1_host[pe["idx"] * sN["cm1_n"] + pe["polId"]]))
        elif pe["polType"] == "imP":                                                      # calculate_exp_at_point :558-572
            if publics:
                _check(L.zk_dev_upload(d_pub.ptr, _ptr(np.array(publics, np.uint64)), 8 * len(publics)))
            run(setup.public_programs[i], "n")
            v = B["scratch"].to_host()[3 * pe["idx"]:3 * pe["idx"] + 3]
            publics.append(int(v[0]))
        else:
            raise ZkError("Invalid public type " + pe["polType"])
    if publics:
        _check(L.zk_dev_upload(d_pub.ptr, _ptr(np.array(publics, np.uint64)), 8 * len(publics)))
    tr = TranscriptGL()
    if publics:
        tr.put_dev(d_pub, len(publics))

    keep = []                                                                             # device buffers the trees borrow

    def extend_and_merkelize(sec):                                                        # stark_gen.rs:709-732
        width = sN[sec + "_n"]
        if width:
            tmp = DevArray(width * Next)
            _check(L.zk_gl_lde_dev(B[sec + "_n"].ptr, width, nbits, B[sec + "_2ns"].ptr, tmp.ptr, nbits_ext, None))
            keep.append(tmp)
        t = MerkleTreeGL()
        t.merkelize_dev(B[sec + "_2ns"].ptr, width, Next)
        return t

    def challenge(i):
        _check(L.zk_transcript_get_field_dev(tr._h, d_chal.ptr + 24 * i, None))

    def put_root(tree):
        tr.put_dev(_Raw(_tree_root_dev(tree), 4))

    tree1 = extend_and_merkelize("cm1"); put_root(tree1)
    challenge(0); challenge(1)                                                            # u, defVal
    run(setup.programs["step2prev"], "n")
    n_cm = info["n_cm1"]
    e2p = lambda k: info["exp2pol"][k] if k in info["exp2pol"] else info["exp2pol"][str(k)]
    for pu in info["pu_ctx"]:                                                             # stark_gen.rs:300-308
        f, t = get_pol(e2p(pu["f_exp_id"])), get_pol(e2p(pu["t_exp_id"]))
        h1, h2 = calculate_h1h2_dev(f, t)
        set_pol(info["cm_n"][n_cm], h1); n_cm += 1
        set_pol(info["cm_n"][n_cm], h2); n_cm += 1
    tree2 = extend_and_merkelize("cm2"); put_root(tree2)
    challenge(2); challenge(3)                                                            # gamma, beta
    _zero(B["tmpexp_n"])                                  # an output-only section starts from zero (stark_gen.rs:944-951)
    run(setup.programs["step3prev"], "n")
    n_cm = info["n_cm1"] + info["n_cm2"]
    for o in info["pu_ctx"] + info["pe_ctx"] + info["ci_ctx"]:                            # stark_gen.rs:329-353
        num, den = get_pol(e2p(o["num_id"])), get_pol(e2p(o["den_id"]))
        z = DevArray(3 * N)
        _check(L.zk_stark_calculate_z_dev(num.ptr, den.ptr, N, z.ptr, None))
        set_pol(info["cm_n"][n_cm], z); n_cm += 1
    _zero(B["tmpexp_n"])
    run(setup.programs["step3"], "n")
    tree3 = extend_and_merkelize("cm3"); put_root(tree3)
    challenge(4)                                                                          # vc
    run(setup.programs["step42ns"], "2ns")
    q_dim, q_deg = info["q_dim"], info["q_deg"]                                           # stark_gen.rs:375-396
    qq1, tmpq = DevArray(q_dim * Next), DevArray(q_dim * Next)
    _check(L.zk_gl_ntt_dev(B["q_2ns"].ptr, qq1.ptr, tmpq.ptr, q_dim, nbits_ext, 1, None))
    if q_deg > 0:
        qq2 = qsplit(qq1, nbits, nbits_ext, q_dim, q_deg)
        tmp4 = DevArray(q_dim * q_deg * Next)
        _check(L.zk_gl_ntt_dev(qq2.ptr, B["cm4_2ns"].ptr, tmp4.ptr, q_dim * q_deg, nbits_ext, 0, None))
        keep += [qq2, tmp4]
    tree4 = MerkleTreeGL(); tree4.merkelize_dev(B["cm4_2ns"].ptr, sN["cm4_2ns"], Next)   # stark_gen.rs:399-405
    put_root(tree4)
    challenge(7)                                                                          # xi
    d_xi = _Raw(d_chal.ptr + 24 * 7, 3)
    LEv, LpEv = lev(d_xi, nbits, False), lev(d_xi, nbits, True)                           # stark_gen.rs:416-430
    descs = []
    for ev in info["ev_map"]:                                                             # stark_gen.rs:432-466
        if ev["type_"] == "const":
            descs.append((B["const_2ns"], info["n_constants"], ev["id"], 1, ev["prime"]))
        elif ev["type_"] == "cm":
            p = info["var_pol_map"][info["cm_2ns"][ev["id"]]]
            descs.append((B[p["section"]], sN[p["section"]], p["section_pos"], p["dim"], ev["prime"]))
        else:
            raise ZkError("Invalid ev type: " + ev["type_"])
    if descs:
        arr = (EvalDesc * len(descs))(*[EvalDesc(b.ptr, w, o, d, int(pr)) for (b, w, o, d, pr) in descs])
        _check(L.zk_stark_evals_dev(arr, len(descs), nbits, ext, LEv.ptr, LpEv.ptr, d_evals.ptr, None))
        tr.put_dev(d_evals, 3 * n_ev)                                                     # stark_gen.rs:469-472
    challenge(5); challenge(6)                                                            # v1, v2
    w = L.zk_gl_root_of_unity(nbits)
    dev["xdiv"], dev["xdivw"] = xdivxsub(d_xi, 1, nbits_ext), xdivxsub(d_xi, w, nbits_ext)  # stark_gen.rs:481-522
    run(setup.programs["step52ns"], "2ns")

    trees0 = [tree1, tree2, tree3, tree4, setup.const_tree]
    fri_proof = fri_prove(tr, B["f_2ns"], ss, lambda idx: [_group_proof(t, idx) for t in trees0])
    ev_host = d_evals.to_host()[:3 * n_ev].reshape(-1, 3)
    return {"rootC": [int(v) for v in setup.const_root], "root1": _ints(tree1.root()), "root2": _ints(tree2.root()),
            "root3": _ints(tree3.root()), "root4": _ints(tree4.root()), "fri_proof": fri_proof,
            "evals": [[int(v) for v in e] for e in ev_host], "publics": publics}


def _ints(a):
    return [int(v) for v in a]


def _zero(d):
    if d.n:
        _check(lib().zk_dev_memset(d.ptr, 0, d.n * 8))


def _group_proof(tree, idx):
    row, path = tree.get_group_proof(idx)
    return [int(v) for v in row], [[int(x) for x in lvl] for lvl in path]


def calculate_h1h2_dev(f, t):
    """calculate_H1H2 (stark_gen.rs:624-651) on [N][3] device polynomials -> (h1, h2), device polynomials"""
    n = f.n // 3
    h1, h2 = DevArray(3 * n), DevArray(3 * n)
    _check(lib().zk_stark_calculate_h1h2_dev(f.ptr, t.ptr, n, h1.ptr, h2.ptr, None))
    return h1, h2


def fri_prove(tr, d_pol, ss, query_pol):
    """FRI::prove (fri.rs:84-184); folding, transposition, Merkle commitments and the transcript on the device."""
    L = lib()
    steps = [s["nBits"] for s in ss["steps"]]
    pol_bits = ss["nBitsExt"]
    shift_inv = pow(49, P - 2, P)
    trees, keep = [], []
    queries = [{"root": None, "pol_queries": []} for _ in steps]
    d_sx = DevArray(3)
    for si, step_bits in enumerate(steps):
        tr.get_field_dev(d_sx)                                                            # special_x
        d_pol = fri_fold(d_pol, pol_bits, step_bits, d_sx, shift_inv)
        if si < len(steps) - 1:
            nxt = steps[si + 1]
            n_groups, group_size = 1 << nxt, (1 << step_bits) >> nxt
            tb = fri_transpose(d_pol, 1 << step_bits, nxt)
            t = MerkleTreeGL(); t.merkelize_dev(tb.ptr, 3 * group_size, n_groups)
            trees.append(t); keep.append(tb)
            queries[si + 1]["root"] = _ints(t.root())
            tr.put_dev(_Raw(_tree_root_dev(t), 4))
        else:
            tr.put_dev(d_pol, 3 << step_bits)                                             # fri.rs:136-141
        for _ in range(pol_bits - step_bits):
            shift_inv = shift_inv * shift_inv % P
        pol_bits = step_bits
    ys = [int(v) for v in tr.get_permutations(ss["nQueries"], steps[0])]                  # fri.rs:158
    for si in range(len(steps)):
        for y in ys:
            queries[si]["pol_queries"].append(query_pol(y) if si == 0 else [_group_proof(trees[si - 1], y)])
        if si < len(steps) - 1:
            ys = [y % (1 << steps[si + 1]) for y in ys]
    last = d_pol.to_host().reshape(-1, 3)
    return {"queries": queries, "last": [[int(v) for v in e] for e in last]}


# ---- serializer.rs:140-264 (zkin.json) ---------------------------------------------------------------
def _digest(d):                                                                            # digest.rs:84-112
    return str(d[0]) if d[1] == 0 and d[2] == 0 and d[3] == 0 else [str(v) for v in d]


def to_zkin(proof):
    z = {"rootC": _digest(proof["rootC"])}
    for k in ("root1", "root2", "root3", "root4"):
        z[k] = _digest(proof[k])
    z["evals"] = [[str(v) for v in e] for e in proof["evals"]]
    qs = proof["fri_proof"]["queries"]
    sib = lambda path: [[str(v) for v in lvl] for lvl in path]
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


def generate_program(pil_json, stark_struct_json):
    """StarkInfo::new (starkinfo.rs:160-272) inside the library: compiled PIL + StarkStruct (JSON text) -> the
    '{"starkinfo": ..., "program": ...}' text NativeStarkSetup takes.  Host only, no GPU needed."""
    import ctypes
    p = lib().zk_starkinfo_generate(pil_json.encode(), stark_struct_json.encode())
    if not p:
        raise ZkError(lib().zk_last_error().decode())
    try:
        return ctypes.string_at(p).decode()
    finally:
        lib().zk_string_free(p)


class NativeStarkSetup:
    """The C++ driver inside libzkgpu (csrc/stark_prover.hip): StarkSetup::new + stark_gen + FRI::prove
    behind zk_stark_setup_new / zk_stark_gen.  `program_json` = '{"starkinfo": ..., "program": ...}' text."""

    def __init__(self, const_n, program_json, stark_struct_json, prover_addr=None, self_check=False):
        """self_check: every gen() verifies its own proof before returning it, as stark_prove does (prove.rs:124-132)"""
        c = _np(const_n)
        hash_type = json.loads(stark_struct_json).get("verificationHashType")
        if hash_type in ("BN128", "BLS12381"):
            from . import bn128_init
            bn128_init(field=hash_type.lower())
        self._h = lib().zk_stark_setup_new(program_json.encode(), stark_struct_json.encode(), _ptr(c), c.size)
        if not self._h:
            raise ZkError(lib().zk_last_error().decode())
        if prover_addr is not None:
            _check(lib().zk_stark_setup_set_prover_addr(self._h, prover_addr.encode()))
        if self_check:
            _check(lib().zk_stark_setup_set_self_check(self._h, 1))

    def verify(self, zkin):
        """stark_verify (stark_verify.rs:20-136) of a proof of this setup: zkin = the dict gen() returned or its JSON text.
        True = accepted, False = rejected (`last_reject()` says which check); malformed input raises ZkError."""
        text = zkin if isinstance(zkin, (str, bytes)) else json.dumps(zkin)
        rc = lib().zk_stark_verify(self._h, text if isinstance(text, bytes) else text.encode())
        if rc < 0:
            raise ZkError(lib().zk_last_error().decode())
        return rc == 1

    @staticmethod
    def last_reject():
        return lib().zk_last_error().decode()

    def const_root(self):
        o = np.zeros(4, np.uint64); _check(lib().zk_stark_setup_const_root(self._h, _ptr(o))); return [int(v) for v in o]

    def setup_timing(self):
        """where StarkSetup::new's time went: {json_parse_ms, const_lde_merkle_ms, programs_ms, hiprtc_compiled, code_cache_*_hits, total_ms}"""
        t = lib().zk_stark_setup_timing(self._h)
        return json.loads(t.decode()) if t else {}

    def last_timing(self):
        """per-stage HIP-event milliseconds of the last proof (reference span names); {} unless ZK_STARK_TIMING=1 was set"""
        t = lib().zk_stark_last_timing(self._h)
        return json.loads(t.decode()) if t else {}

    def gen(self, cm_n, stream=None):
        """-> the proof as the zkin dict of serializer.rs:146-261; cm_n: host array or DevArray (HBM-resident trace).
        stream: a HIP stream handle (int) to prove on; setups on different streams may prove concurrently from different
        host threads (zk_stark_gen_dev_on), one proof at a time per setup."""
        return json.loads(self.gen_json(cm_n, stream))

    def gen_json(self, cm_n, stream=None):
        """the same proof as the JSON text the library wrote (what `zkit stark_prove --o` stores), unparsed"""
        return self.gen_bytes(cm_n, stream).decode()

    def gen_bytes(self, cm_n, stream=None):
        """the proof's JSON text as bytes (no decoding: a caller that stores it or picks a root out of it needs none)"""
        import ctypes
        if isinstance(cm_n, DevArray):
            p = lib().zk_stark_gen_dev_on(self._h, cm_n.ptr, cm_n.n, stream)
        else:
            if stream:
                raise ZkError("gen: a stream needs an HBM-resident trace (DevArray); zk_stark_gen takes host traces on the null stream only")
            c = _np(cm_n)
            p = lib().zk_stark_gen(self._h, _ptr(c), c.size)
        if not p:
            raise ZkError(lib().zk_last_error().decode())
        try:
            return ctypes.string_at(p)
        finally:
            lib().zk_string_free(p)

    def staged(self, cm_n, stream=None):
        """one proof in progress, stage by stage (zk_stark_new ...): the seams a caller with its own stark_gen.rs binds"""
        return StagedProof(self, cm_n, stream)

    def free(self):
        if self._h:
            lib().zk_stark_setup_free(self._h); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


STEP_2PREV, STEP_3PREV, STEP_3, STEP_42NS, STEP_52NS = range(5)                # zkgpu.h ZK_STEP_*


class StagedProof:
    """stark_gen cut at the reference's seams (include/zkgpu.h "the staged prover"): `commit_stage` = extend_and_merkelize + transcript.put
    (stark_gen.rs:709-750), `eval` = calculate_exps_parallel (:786-792), `challenge` = transcript.get_field into ctx.challenges, `evals`
    (:416-472), `fri_prove` (fri.rs:84-184), `finish` = openings + zkin.  `run_all()` is the reference's order; its zkin equals gen()'s."""

    def __init__(self, setup, cm_n, stream=None):
        self._setup = setup                                                          # (keeps the setup alive)
        self._cm = None
        if isinstance(cm_n, DevArray):
            # the context BORROWS d_cm_pols until zk_stark_free (zkgpu.h): the step programs, evals() and an early stage 3 read it after
            # this constructor returns -- keep the array (a temporary's __del__ would hand the block back to the pool mid-proof)
            self._cm = cm_n
            self._h = lib().zk_stark_new(setup._h, None, cm_n.ptr, cm_n.n, stream)
        else:
            c = _np(cm_n)
            self._h = lib().zk_stark_new(setup._h, _ptr(c), None, c.size, stream)
        if not self._h:
            raise ZkError(lib().zk_last_error().decode())

    def commit_stage(self, stage):
        o = np.zeros(4, np.uint64); _check(lib().zk_stark_commit_stage(self._h, stage, _ptr(o))); return [int(v) for v in o]

    def challenge(self, i):
        o = np.zeros(3, np.uint64); _check(lib().zk_stark_challenge(self._h, i, _ptr(o))); return [int(v) for v in o]

    def set_challenge(self, i, v):
        a = np.ascontiguousarray(np.array([int(x) for x in v], dtype=np.uint64)); _check(lib().zk_stark_set_challenge(self._h, i, _ptr(a)))

    def eval(self, step): _check(lib().zk_stark_eval(self._h, step))
    def calculate_h1h2(self): _check(lib().zk_stark_calculate_h1h2(self._h))
    def calculate_z(self): _check(lib().zk_stark_calculate_z(self._h))

    def evals(self):
        n = lib().zk_stark_evals(self._h, None, 0)
        if n < 0:
            raise ZkError(lib().zk_last_error().decode())
        return n

    def fri_prove(self): _check(lib().zk_stark_fri_prove(self._h))

    def fri_pol_dev(self): return lib().zk_stark_fri_pol_dev(self._h)
    def tree(self, j): return lib().zk_stark_tree(self._h, j)

    def finish(self):
        import ctypes
        p = lib().zk_stark_finish(self._h)
        if not p:
            raise ZkError(lib().zk_last_error().decode())
        try:
            return ctypes.string_at(p)
        finally:
            lib().zk_string_free(p)

    def run_all(self):
        """stark_gen.rs:279-545, in order -> zkin bytes"""
        self.commit_stage(1)
        return self.run_all_from_stage1()

    def run_all_from_stage1(self):
        """the rest of run_all() once commit_stage(1) has been called"""
        self.challenge(0); self.challenge(1)
        self.eval(STEP_2PREV); self.calculate_h1h2()
        self.commit_stage(2); self.challenge(2); self.challenge(3)
        self.eval(STEP_3PREV); self.calculate_z(); self.eval(STEP_3)
        self.commit_stage(3); self.challenge(4)
        self.eval(STEP_42NS)
        self.commit_stage(4); self.challenge(7)
        self.evals(); self.challenge(5); self.challenge(6)
        self.eval(STEP_52NS)
        self.fri_prove()
        return self.finish()

    def free(self):
        if self._h:
            lib().zk_stark_free(self._h); self._h = None
        self._cm = None                                                              # only now may the trace go back to the pool

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def fri_prove_dev(transcript, d_pol, nbits_ext, steps, n_queries, query_trees, stream=None):
    """FRI::prove(transcript, pol, query_pol) (fri.rs:84-184) alone: transcript = a TranscriptGL handle of this library (zk_transcript_t*),
    d_pol = device pointer to 3 << nbits_ext words, query_trees = zk_merkle_t* handles -> the FRI part of a zkin as a dict (+ "ys")"""
    import ctypes
    st = (ctypes.c_uint32 * len(steps))(*steps)
    tr = (ctypes.c_void_p * max(1, len(query_trees)))(*query_trees)
    p = lib().zk_fri_prove_dev(transcript, d_pol, nbits_ext, st, len(steps), n_queries, tr, len(query_trees), stream)
    if not p:
        raise ZkError(lib().zk_last_error().decode())
    try:
        return json.loads(ctypes.string_at(p))
    finally:
        lib().zk_string_free(p)


class reference_compat_paths:
    """`with reference_compat_paths():` -- inside, 16-ary Merkle paths are checked as loosely as merklehash_bn128.rs:108-128 checks them
    (only the last level bound to the root).  The library's default is strict (every level linked, zkgpu.h); this switch exists for parity
    tests against a verifier that follows the reference to the letter.  Process-wide."""
    def __enter__(self):
        self._old = lib().zk_stark_verify_set_reference_compat(1)
        return self
    def __exit__(self, *a):
        lib().zk_stark_verify_set_reference_compat(self._old)


def stark_verify(zkin, const_root, program_json, stark_struct_json):
    """stark_verify without a prover's setup (zk_stark_verify_with): const_root = 4 words (GL) or the raw limbs
    NativeStarkSetup.const_root() returns for scalar-field hashing"""
    hash_type = json.loads(stark_struct_json).get("verificationHashType")
    if hash_type in ("BN128", "BLS12381"):
        from . import bn128_init
        bn128_init(field=hash_type.lower())
    r = np.ascontiguousarray(np.array([int(v) for v in const_root], dtype=np.uint64))
    text = zkin if isinstance(zkin, (str, bytes)) else json.dumps(zkin)
    rc = lib().zk_stark_verify_with(program_json.encode(), stark_struct_json.encode(), _ptr(r), text if isinstance(text, bytes) else text.encode())
    if rc < 0:
        raise ZkError(lib().zk_last_error().decode())
    return rc == 1


def load_program_json(path):
    """{"starkinfo": ..., "program": ...} as serialised by the reference (serde) or by a generator."""
    d = json.load(open(path))
    info = d["starkinfo"]
    info["exp2pol"] = {int(k): v for k, v in info["exp2pol"].items()}
    return info, d["program"]


def stark_prove(starkinfo_json, const_path, cm_path, stark_struct_path):
    """starky::prove::stark_prove (prove.rs:30-91) for verificationHashType == "GL"."""
    info, program = load_program_json(starkinfo_json)
    ss = json.load(open(stark_struct_path))
    const_n = np.fromfile(const_path, dtype="<u8")                                        # polsarray.rs:137-217
    cm_n = np.fromfile(cm_path, dtype="<u8")
    setup = StarkSetup(const_n, info, program, ss)
    return setup, stark_gen(cm_n, setup)
