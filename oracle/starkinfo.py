"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Python restatement of the reference's PIL -> prover-program
code generator (pure CPU, runs once per PIL): starky/src/starkinfo.rs:160-408,
starkinfo_codegen.rs:296-669, starkinfo_Z.rs, starkinfo_cp_prover.rs, starkinfo_cp_ver.rs,
starkinfo_fri_prover.rs, starkinfo_fri_ver.rs, starkinfo_map.rs, expressionops.rs.

Output = (starkinfo, program) as plain dicts whose keys are the reference's serde field names
(StarkInfo `starkinfo.rs:46-95`, Program `:27-37`, Segment/Section/Node
`starkinfo_codegen.rs:50-89`), i.e. the JSON a Rust caller would hand to the product after
`serde_json::to_string(&starkinfo)` / `(&program)`.  The product (eigen-zkvm_amd/) never imports
this module: tests write the JSON to disk and the product reads it, as it would from the reference.
"""
import copy

P = 0xFFFFFFFF00000001
CHALLENGE_MAP = {"u": 0, "defVal": 1, "gamma": 2, "beta": 3, "vc": 4, "vf1": 5, "vf2": 6, "xi": 7}  # constant.rs:39-50
GLOBAL_L1 = "Global.L1"                                                                              # constant.rs:118
SECTIONS = ["cm1_n", "cm1_2ns", "cm2_n", "cm2_2ns", "cm3_n", "cm3_2ns", "cm4_n", "cm4_2ns", "q_2ns", "f_2ns", "tmpexp_n"]


# ---- expressionops.rs ----------------------------------------------------------------------------
def E(op, id=None, value=None, values=None, next=None):
    return {"op": op, "deg": 0, "id": id, "next": next, "value": value, "values": values,
            "keep": None, "keep2ns": None, "idQ": None, "const_": None}


def e_add(a, b): return E("add", values=[copy.deepcopy(a), copy.deepcopy(b)])
def e_sub(a, b): return E("sub", values=[copy.deepcopy(a), copy.deepcopy(b)])
def e_mul(a, b): return E("mul", values=[copy.deepcopy(a), copy.deepcopy(b)])
def e_exp(i, next=None): return E("exp", id=i, next=next)
def e_cm(i, next=None): return E("cm", id=i, next=next)
def e_const(i, next=None): return E("const", id=i, next=next)
def e_q(i, next=None): return E("q", id=i, next=next)
def e_challenge(name): return E("challenge", id=CHALLENGE_MAP[name])
def e_number(n): return E("number", value=str(n))
def e_eval(n): return E("eval", id=n)
def e_nop(): return E("nop")
def is_nop(e): return e["op"] == "nop"
def e_next(e): return bool(e.get("next"))


def load_expr(d):
    """normalise a pil.json expression (types.rs:36-60) into the dict shape above"""
    e = E(d["op"], id=d.get("id"), value=d.get("value"), next=d.get("next"))
    e["deg"] = d.get("deg", 0)
    e["const_"] = d.get("const")
    if "const_" in d:
        e["const_"] = d["const_"]
    for k in ("keep", "keep2ns", "idQ"):
        e[k] = d.get(k)
    if d.get("values") is not None:
        e["values"] = [load_expr(v) for v in d["values"]]
    return e


def load_pil(d):
    pil = copy.deepcopy(d)
    pil["expressions"] = [load_expr(e) for e in d["expressions"]]
    pil.setdefault("permutationIdentities", None)
    pil.setdefault("connectionIdentities", None)
    pil["cm_dims"] = []
    pil["q2exp"] = []
    return pil


# ---- starkinfo_codegen.rs ------------------------------------------------------------------------
def Node(type_, id=0, value=None, dim=0, prime=False, tree_pos=0):
    return {"type_": type_, "id": id, "value": value, "dim": dim, "prime": prime, "tree_pos": tree_pos, "p": 0, "exp_id": 0}


def Section(op, dest, src):
    return {"op": op, "dest": dest, "src": src}


class Ctx:  # Context (starkinfo_codegen.rs:15-21)
    def __init__(self):
        self.tmp_used = 0
        self.code = []
        self.calculated = {}


def eval_single_op(cc, exp, prime, values):  # :442-555
    op = exp["op"]
    def tmp():
        r = Node("tmp", cc["tmp_used"]); cc["tmp_used"] += 1; return r
    if op in ("add", "sub", "mul", "muladd"):
        r = tmp(); cc["code"].append(Section(op, copy.deepcopy(r), list(values))); return r
    if op in ("addc", "mulc"):
        a = values[0]; b = Node("number", 0, str(exp["const_"]))
        r = tmp(); cc["code"].append(Section("add" if op == "addc" else "mul", copy.deepcopy(r), [a, b])); return r
    if op == "neg":
        a = Node("number", 0, "0"); b = values[0]
        r = tmp(); cc["code"].append(Section("sub", copy.deepcopy(r), [a, b])); return r
    if op in ("cm", "const", "exp", "q"):
        if e_next(exp) and prime:
            raise ValueError("Double Prime")
        return Node(op, exp["id"], None, 0, e_next(exp) or prime)
    if op == "number":
        return Node("number", 0, exp["value"])
    if op in ("public", "challenge", "eval"):
        return Node(op, exp["id"])
    if op in ("xDivXSubXi", "xDivXSubWXi", "x"):
        return Node(op, 0)
    raise ValueError("InvalidOperator: eval_exp: " + op)


def eval_exp(cc, exp, prime):  # :421-440, left-to-right post-order
    if is_nop(exp):
        raise ValueError("nop expression")
    vals = [eval_exp(cc, v, prime) for v in (exp["values"] or [])]
    return eval_single_op(cc, exp, prime, vals)


def find_muladd(exp):  # :358-384
    if exp["values"] is not None:
        v = exp["values"]
        if exp["op"] == "add" and v[0]["op"] == "mul":
            vv = v[0]["values"]
            return E("muladd", values=[find_muladd(vv[0]), find_muladd(vv[1]), find_muladd(v[1])])
        if exp["op"] == "add" and v[1]["op"] == "mul":
            vv = v[1]["values"]
            return E("muladd", values=[find_muladd(vv[0]), find_muladd(vv[1]), find_muladd(v[0])])
        r = copy.deepcopy(exp)
        mv = [find_muladd(x) for x in v]
        if mv:
            r["values"] = mv
        return r
    return copy.deepcopy(exp)


def calculate_deps(ctx, pil, expr, prime, exp_id, muladd):  # :557-578
    if expr["op"] == "exp":
        if prime and e_next(expr):
            raise ValueError("Double prime")
        pil_code_gen(ctx, pil, expr["id"], prime or e_next(expr), "", 0, muladd)
    for e in (expr["values"] or []):
        calculate_deps(ctx, pil, e, prime, exp_id, muladd)


def pil_code_gen(ctx, pil, exp_id, prime, res_type, res_id, muladd):  # :296-356
    key = ("expsPrime" if prime else "exps", exp_id)
    if key in ctx.calculated:                                             # contains_key, whatever the value
        if res_type:
            c = next(x for x in ctx.code if x["exp_id"] == exp_id and x["prime"] == prime)
            dest = Node(res_type, res_id, None, 0, prime)
            c["code"].append(Section("copy", dest, [copy.deepcopy(c["code"][-1]["dest"])]))
        return
    exp = copy.deepcopy(pil["expressions"][exp_id])
    calculate_deps(ctx, pil, exp, prime, exp_id, False)
    cc = {"exp_id": exp_id, "tmp_used": ctx.tmp_used, "code": []}
    _exp = copy.deepcopy(pil["expressions"][exp_id])
    exp = find_muladd(_exp) if muladd else _exp
    ret = eval_exp(cc, exp, prime)
    if ret["type_"] == "tmp":
        cc["code"][-1]["dest"] = Node("exp", exp_id, None, 0, prime)
        cc["tmp_used"] -= 1
    else:
        cc["code"].append(Section("copy", Node("exp", exp_id, None, 0, prime), [ret]))
    if res_type:
        if prime:
            raise ValueError("Prime in retType")
        cc["code"].append(Section("copy", Node(res_type, res_id, None, 0, prime), [Node("exp", exp_id, None, 0, prime)]))
    ctx.code.append({"exp_id": exp_id, "prime": prime, "code": cc["code"], "tmp_used": 0, "idQ": None})
    ctx.calculated[key] = True
    if cc["tmp_used"] > ctx.tmp_used:
        ctx.tmp_used = cc["tmp_used"]


def _exp_and_expprimes(ctx, pil):  # :635-658
    calc = {}
    for c in ctx.code:
        e = pil["expressions"][c["exp_id"]]
        if e["idQ"] is not None or e["keep"] is not None or e["keep2ns"] is not None:
            calc[c["exp_id"]] = calc.get(c["exp_id"], 0) | (2 if c["prime"] else 1)
    return {k: v == 3 for k, v in calc.items()}


def build_linear_code(ctx, pil, loop_pos):  # :610-632
    ep = _exp_and_expprimes(ctx, pil) if loop_pos in ("i", "last") else {}
    res = []
    for i, c in enumerate(ctx.code):
        if ep.get(i) and ((loop_pos == "i" and not c["prime"]) or loop_pos == "last"):
            continue
        res.extend(copy.deepcopy(c["code"]))
    return res


def build_code(ctx, pil):  # :586-603
    seg = {"first": build_linear_code(ctx, pil, "first"), "i": build_linear_code(ctx, pil, "i"),
           "last": build_linear_code(ctx, pil, "last"), "tmp_used": ctx.tmp_used}
    for i, e in enumerate(pil["expressions"]):
        if e["keep"] is None and e["idQ"] is None:
            ctx.calculated[("exps", i)] = False
            ctx.calculated[("expsPrime", i)] = False
    ctx.code = []
    return seg


def iterate_code(seg, f):  # :660-669
    for part in ("first", "i", "last"):
        for c in seg[part]:
            for s in c["src"]:
                f(s)
            f(c["dest"])


# ---- degree (starkinfo_cp_prover.rs:243-270) and dimension (starkinfo_map.rs:619-640) of an expression
def exp_degree(pil, exp):
    op, v = exp["op"], exp["values"] or []
    if op in ("add", "sub", "addc", "mulc", "neg"):
        return max([1] + [exp_degree(pil, x) for x in v])
    if op == "mul":
        return exp_degree(pil, v[0]) + exp_degree(pil, v[1])
    if op == "muladd":
        return max(exp_degree(pil, v[0]) + exp_degree(pil, v[1]), exp_degree(pil, v[2]))
    if op in ("cm", "const", "x"):
        return 1
    if op == "exp":
        return exp_degree(pil, pil["expressions"][exp["id"]])
    if op in ("number", "public", "challenge", "eval"):
        return 0
    raise ValueError("Exp op not defined: " + op)


def exp_dim(pil, exp):
    op = exp["op"]
    if op in ("add", "sub", "mul", "muladd", "addc", "mulc", "neg"):
        return max([1] + [exp_dim(pil, x) for x in exp["values"]])
    if op == "cm":
        return pil["cm_dims"][exp["id"]]
    if op == "exp":
        return exp_dim(pil, pil["expressions"][exp["id"]])
    if op == "q":
        return exp_dim(pil, pil["expressions"][pil["q2exp"][exp["id"]]])
    if op in ("const", "number", "public", "x"):
        return 1
    if op in ("challenge", "eval", "xDivXSubXi", "xDivXSubWXi"):
        return 3
    raise ValueError("Exp op not defined: " + op)


# ---- calculate_im_pols (starkinfo_cp_prover.rs:123-291) -------------------------------------------
def _calc_im(pil, exp, im, max_deg, abs_max, st):
    if im is None:
        return None, -1
    op = exp["op"]
    if op in ("add", "sub", "addc", "mulc", "neg"):
        md, im_e = 0, dict(im)
        for v in exp["values"]:
            im_e, d = _calc_im(pil, v, im_e, max_deg, abs_max, st)
            md = max(md, d)
        return im_e, md
    if op in ("number", "public", "challenge"):
        return dict(im), 0
    if op in ("x", "const", "cm"):
        return (None, -1) if max_deg < 1 else (dict(im), 1)
    if op == "mul":
        v = exp["values"]
        if v[0]["op"] in ("number", "public", "challenge"):
            return _calc_im(pil, v[1], im, max_deg, abs_max, st)
        if v[1]["op"] in ("number", "public", "challenge"):
            return _calc_im(pil, v[0], im, max_deg, abs_max, st)
        here = exp_degree(pil, exp)
        if here <= max_deg:
            return dict(im), here
        eb, ed = None, -1
        for l in range(max_deg + 1):
            e1, d1 = _calc_im(pil, v[0], im, l, abs_max, st)
            e2, d2 = _calc_im(pil, v[1], e1, max_deg - l, abs_max, st)
            if e2 is not None and (eb is None or len(e2) < len(eb)):
                eb, ed = e2, d1 + d2
            if eb is not None and len(eb) == len(im):
                return eb, ed
        return eb, ed
    if op == "exp":
        if max_deg < 1:
            return None, -1
        if exp["id"] in im:
            return dict(im), 1
        e, d = _calc_im(pil, pil["expressions"][exp["id"]], im, abs_max, abs_max, st)
        if e is None:
            return None, -1
        if d > max_deg:
            e[exp["id"]] = True
            st[0] = max(st[0], d)
            return e, 1
        return e, d
    raise ValueError("Exp op not defined: " + op)


def calculate_im_pols(pil, exp, max_deg):
    st = [0]
    re, rd = _calc_im(pil, exp, {}, max_deg, max_deg, st)
    return re, max(rd, st[0]) - 1


def get_ks(n):  # helper.rs:16-23
    ks = [12275445934081160404]
    for _ in range(1, n):
        ks.append(ks[-1] * ks[0] % P)
    return ks[:n]


# ---- StarkInfo::new (starkinfo.rs:160-272) --------------------------------------------------------
def PCCTX():
    return {"f_exp_id": 0, "t_exp_id": 0, "h1_id": 0, "h2_id": 0, "z_id": 0, "c1_id": 0, "c2_id": 0, "num_id": 0, "den_id": 0}


def _lc(pil, ids, u, left_mul):
    """t = Horner in u over exps `ids`; left_mul: u*acc (t side) vs acc*u (f side)"""
    acc = e_nop()
    for j in ids:
        e = e_exp(j)
        acc = e if is_nop(acc) else e_add(e_mul(u, acc) if left_mul else e_mul(acc, u), e)
    return acc


def generate(pil_json, stark_struct, global_l1=None):
    pil = load_pil(pil_json)
    pil_deg = next(iter(pil["references"].values()))["polDeg"]
    if (1 << stark_struct["nBits"]) != pil_deg:
        raise ValueError("stark_deg != pil_deg")
    if stark_struct["nBitsExt"] != stark_struct["steps"][0]["nBits"]:
        raise ValueError("MustEqualDegreeError: stark_struct.nBitsExt != stark_struct.steps[0].nBits")
    info = {"var_pol_map": [], "n_cm1": 0, "n_cm2": 0, "n_cm3": 0, "n_cm4": 0, "n_q": 0, "pu_ctx": [], "pe_ctx": [],
            "ci_ctx": [], "n_constants": pil["nConstants"], "n_publics": len(pil["publics"]), "c_exp": 0, "im_exps": {},
            "q_deg": 0, "q_dim": 0, "im_exps_list": [], "im_exp2cm": {}, "qs": [], "exps_2ns": [], "exps_n": [],
            "ev_map": [], "fri_exp_id": 0, "n_exps": 0, "cm_n": [], "cm_2ns": [], "tmpexp_n": [], "q_2ns": [], "f_2ns": [],
            "map_sections": {s: [] for s in SECTIONS}, "map_sectionsN1": {s: 0 for s in SECTIONS},
            "map_sectionsN3": {s: 0 for s in SECTIONS}, "map_sectionsN": {s: 0 for s in SECTIONS},
            "map_offsets": {s: 0 for s in SECTIONS}, "map_deg": {s: 0 for s in SECTIONS}, "map_total_n": 0,
            "exp2pol": {}, "publics": [], "ev_idx": {"cm": {}, "const_": {}}}
    prog = {"publics_code": [], "step2prev": None, "step3prev": None, "step3": None, "step42ns": None, "step52ns": None,
            "verifier_code": None, "verifier_query_code": None}
    l1_name = global_l1 or GLOBAL_L1

    # -- generate_public_calculators (:274-322)
    for p in pil["publics"]:
        if p["polType"] == "imP":
            ctx = Ctx()
            pil_code_gen(ctx, pil, p["polId"], False, "", 0, False)
            seg = build_code(ctx, pil)
            st = {"map": {}, "tmp_used": seg["tmp_used"]}
            def fix(r, st=st):
                if r["type_"] == "exp":
                    k = (1 if r["prime"] else 0, r["id"])
                    if k not in st["map"]:
                        st["map"][k] = st["tmp_used"]; st["tmp_used"] += 1
                    r["prime"] = False; r["type_"] = "tmp"; r["id"] = st["map"][k]
            iterate_code(seg, fix)
            seg["tmp_used"] = st["tmp_used"]
            prog["publics_code"].append(seg)
    info["n_cm1"] = pil["nCommitments"]
    ctx, ctx2ns = Ctx(), Ctx()

    # -- generate_step2 (:324-408): plookup h1, h2
    u, def_val = e_challenge("u"), e_challenge("defVal")
    for pi in pil["plookupIdentities"]:
        t_exp = _lc(pil, pi["t"], u, True)
        if pi.get("selT") is not None:
            t_exp = e_add(e_mul(e_sub(t_exp, def_val), e_exp(pi["selT"])), def_val)
            t_exp["idQ"] = pil["nQ"]; pil["nQ"] += 1
        t_exp_id = len(pil["expressions"]); t_exp["keep"] = True; pil["expressions"].append(t_exp)
        f_exp = _lc(pil, pi["f"], u, False)
        if pi.get("selF") is not None:
            f_exp = e_add(e_mul(e_sub(f_exp, e_exp(t_exp_id)), e_exp(pi["selF"])), e_exp(t_exp_id))
            f_exp["idQ"] = pil["nQ"]; pil["nQ"] += 1
        f_exp_id = len(pil["expressions"]); f_exp["keep"] = True; pil["expressions"].append(f_exp)
        pil_code_gen(ctx, pil, f_exp_id, False, "", 0, False)
        pil_code_gen(ctx, pil, t_exp_id, False, "", 0, False)
        c = PCCTX(); c.update(f_exp_id=f_exp_id, t_exp_id=t_exp_id, h1_id=pil["nCommitments"], h2_id=pil["nCommitments"] + 1)
        pil["nCommitments"] += 2
        info["pu_ctx"].append(c)
    prog["step2prev"] = build_code(ctx, pil)
    ctx.calculated.clear()
    info["n_cm2"] = pil["nCommitments"] - info["n_cm1"]

    # -- generate_step3 (starkinfo_Z.rs)
    gamma, beta = e_challenge("gamma"), e_challenge("beta")
    one = e_number(1)
    def l1_const():
        if l1_name not in pil["references"]:
            raise ValueError(l1_name + " must be defined")
        return e_const(pil["references"][l1_name]["id"])
    def push_identity(e):
        e["deg"] = 2
        i = len(pil["expressions"]); pil["expressions"].append(e)
        pil["polIdentities"].append({"e": i, "line": 0, "fileName": ""})
        return i
    for pi in (pil["permutationIdentities"] or []):                       # generate_permutation_LC :32-105
        t_exp = _lc(pil, pi["t"], u, True)
        if pi.get("selT") is not None:
            t_exp = e_add(e_mul(e_sub(t_exp, def_val), e_exp(pi["selT"])), def_val)
            t_exp["idQ"] = pil["nQ"]; pil["nQ"] += 1
        t_exp_id = len(pil["expressions"]); pil["expressions"].append(t_exp)
        f_exp = _lc(pil, pi["f"], u, False)
        if pi.get("selF") is not None:
            f_exp = e_add(e_mul(e_sub(f_exp, def_val), e_exp(pi["selF"])), def_val)
            f_exp["idQ"] = pil["nQ"]; pil["nQ"] += 1
        f_exp_id = len(pil["expressions"]); pil["expressions"].append(f_exp)
        c = PCCTX(); c.update(f_exp_id=f_exp_id, t_exp_id=t_exp_id); info["pe_ctx"].append(c)
    for i in range(len(pil["plookupIdentities"])):                        # generate_plookup_Z :108-199
        pu = info["pu_ctx"][i]
        pu["z_id"] = pil["nCommitments"]; pil["nCommitments"] += 1
        h1, h2, h1p = e_cm(pu["h1_id"]), e_cm(pu["h2_id"]), e_cm(pu["h1_id"], True)
        f, t, tp = e_exp(pu["f_exp_id"]), e_exp(pu["t_exp_id"]), e_exp(pu["t_exp_id"], True)
        z, zp = e_cm(pu["z_id"]), e_cm(pu["z_id"], True)
        pu["c1_id"] = push_identity(e_mul(l1_const(), e_sub(z, one)))
        g1b = e_mul(gamma, e_add(one, beta))
        num = e_mul(e_mul(e_add(f, gamma), e_add(e_add(t, e_mul(tp, beta)), g1b)), e_add(one, beta))
        num["idQ"] = pil["nQ"]; pil["nQ"] += 1; num["keep"] = True
        pu["num_id"] = len(pil["expressions"]); pil["expressions"].append(num)
        den = e_mul(e_add(e_add(h1, e_mul(h2, beta)), g1b), e_add(e_add(h2, e_mul(h1p, beta)), g1b))
        den["idQ"] = pil["nQ"]; pil["nQ"] += 1
        pu["den_id"] = len(pil["expressions"]); den["keep"] = True; pil["expressions"].append(den)
        pu["c2_id"] = push_identity(e_sub(e_mul(zp, e_exp(pu["den_id"])), e_mul(z, e_exp(pu["num_id"]))))
        pil_code_gen(ctx, pil, pu["num_id"], False, "", 0, False)
        pil_code_gen(ctx, pil, pu["den_id"], False, "", 0, False)
    for i in range(len(pil["permutationIdentities"] or [])):              # generate_permutation_Z :201-271
        pe = info["pe_ctx"][i]
        pe["z_id"] = pil["nCommitments"]; pil["nCommitments"] += 1
        f, t, z, zp = e_exp(pe["f_exp_id"]), e_exp(pe["t_exp_id"]), e_cm(pe["z_id"]), e_cm(pe["z_id"], True)
        pe["c1_id"] = push_identity(e_mul(l1_const(), e_sub(z, one)))
        num = e_add(f, beta); num["keep"] = True
        pe["num_id"] = len(pil["expressions"]); pil["expressions"].append(num)
        den = e_add(t, beta); den["keep"] = True
        pe["den_id"] = len(pil["expressions"]); pil["expressions"].append(den)
        pe["c2_id"] = push_identity(e_sub(e_mul(zp, e_exp(pe["den_id"])), e_mul(z, e_exp(pe["num_id"]))))
        pil_code_gen(ctx, pil, pe["num_id"], False, "", 0, False)
        pil_code_gen(ctx, pil, pe["den_id"], False, "", 0, False)
    for ci in (pil["connectionIdentities"] or []):                        # generate_connections_Z :273-423
        pols, conns = ci["pols"], ci["connections"]
        c = PCCTX(); c["z_id"] = pil["nCommitments"]; pil["nCommitments"] += 1
        num = e_add(e_add(e_exp(pols[0]), e_mul(beta, E("x"))), gamma); num["keep"] = True
        den = e_add(e_add(e_exp(pols[0]), e_mul(beta, e_exp(conns[0]))), gamma); den["keep"] = True
        c["num_id"] = len(pil["expressions"]); pil["expressions"].append(num)
        c["den_id"] = len(pil["expressions"]); pil["expressions"].append(den)
        ks = get_ks(len(pols) - 1)
        for i in range(1, len(pols)):
            num = e_mul(e_exp(c["num_id"]), e_add(e_add(e_exp(pols[i]), e_mul(e_mul(beta, e_number(ks[i - 1])), E("x"))), gamma))
            num["idQ"] = pil["nQ"]; pil["nQ"] += 1
            den = e_mul(e_exp(c["den_id"]), e_add(e_add(e_exp(pols[i]), e_mul(beta, e_exp(conns[i]))), gamma))
            den["idQ"] = pil["nQ"]; pil["nQ"] += 1
            c["num_id"] = len(pil["expressions"]); pil["expressions"].append(num)
            c["den_id"] = len(pil["expressions"]); pil["expressions"].append(den)
        z, zp = e_cm(c["z_id"]), e_cm(c["z_id"], True)
        c["c1_id"] = push_identity(e_mul(l1_const(), e_sub(z, one)))
        c["c2_id"] = push_identity(e_sub(e_mul(zp, e_exp(c["den_id"])), e_mul(z, e_exp(c["num_id"]))))
        pil_code_gen(ctx, pil, c["num_id"], False, "", 0, False)
        pil_code_gen(ctx, pil, c["den_id"], False, "", 0, False)
        info["ci_ctx"].append(c)
    prog["step3prev"] = build_code(ctx, pil)
    ctx.calculated.clear()

    # -- generate_constraint_polynomial (starkinfo_cp_prover.rs:12-119)
    vc = e_challenge("vc")
    c_exp = e_nop()
    for pi in pil["polIdentities"]:
        e = e_exp(pi["e"])
        c_exp = e if is_nop(c_exp) else e_add(e_mul(vc, c_exp), e)
    max_deg = (1 << (stark_struct["nBitsExt"] - stark_struct["nBits"])) + 1
    for d in range(2, max_deg + 1):
        im, qd = calculate_im_pols(pil, c_exp, d)
        if im is not None and (info["q_deg"] == 0 or len(im) + qd < len(info["im_exps"]) + info["q_deg"]):
            info["q_deg"], info["im_exps"] = qd, im
    info["im_exps_list"] = sorted(info["im_exps"].keys())
    for k in info["im_exps_list"]:
        info["im_exp2cm"][k] = pil["nCommitments"]; pil["nCommitments"] += 1
        e = E("sub", values=[copy.deepcopy(pil["expressions"][k]), E("cm", id=pil["nCommitments"] - 1)])
        c_exp = e if is_nop(c_exp) else e_add(e_mul(vc, c_exp), e)
    info["c_exp"] = len(pil["expressions"]); pil["expressions"].append(c_exp)
    info["n_cm3"] = pil["nCommitments"] - info["n_cm1"] - info["n_cm2"]
    info["qs"] = []
    for _ in range(info["q_deg"]):
        info["qs"].append(pil["nCommitments"]); pil["nCommitments"] += 1
    for k in info["im_exps_list"]:
        pil_code_gen(ctx, pil, k, False, "", 0, False)
    prog["step3"] = build_code(ctx, pil)
    for k, v in info["im_exps"].items():
        ctx2ns.calculated[("exps", k)] = v; ctx2ns.calculated[("expsPrime", k)] = v
    pil_code_gen(ctx2ns, pil, info["c_exp"], False, "", 0, False)
    code = ctx2ns.code[-1]["code"]
    code.append(Section("mul", Node("q", 0), [copy.deepcopy(code[-1]["dest"]), Node("Zi", 0)]))
    prog["step42ns"] = build_code(ctx2ns, pil)
    info["n_cm4"] = info["q_deg"]

    # -- generate_constraint_polynomial_verifier (starkinfo_cp_ver.rs:8-119)
    ctx = Ctx()
    for k, v in info["im_exps"].items():
        ctx.calculated[("exps", k)] = v; ctx.calculated[("expsPrime", k)] = v
    pil_code_gen(ctx, pil, info["c_exp"], False, "", 0, True)
    code = build_code(ctx, pil)
    st = {"map": {}, "tmp_used": code["tmp_used"]}
    def ev_index(type_, p, id_, prime):
        m = info["ev_idx"]["cm" if type_ == "cm" else "const_"]
        if (p, id_) not in m:
            m[(p, id_)] = len(info["ev_map"])
            info["ev_map"].append(Node(type_, id_, None, 0, prime))
        return m[(p, id_)]
    def fix_ver(r):
        p = 1 if r["prime"] else 0
        t = r["type_"]
        if t == "exp":
            if r["id"] in info["im_exps_list"]:
                r["type_"] = "cm"; r["id"] = info["im_exp2cm"][r["id"]]
                idx = ev_index("cm", p, r["id"], r["prime"])
                r["prime"] = False; r["id"] = idx; r["type_"] = "eval"
            else:
                k = (p, r["id"])
                if k not in st["map"]:
                    st["map"][k] = st["tmp_used"]; st["tmp_used"] += 1
                r["type_"] = "tmp"; r["exp_id"] = r["id"]; r["id"] = st["map"][k]
        elif t in ("cm", "const"):
            idx = ev_index(t, p, r["id"], r["prime"])
            r["prime"] = False; r["id"] = idx; r["type_"] = "eval"
        elif t not in ("number", "challenge", "public", "tmp", "Z", "x", "eval"):
            raise ValueError("Invalid reference type: %r" % r)
    iterate_code(code, fix_ver)
    for i in range(info["q_deg"]):
        info["ev_idx"]["cm"][(0, info["qs"][i])] = len(info["ev_map"])
        info["ev_map"].append(Node("cm", info["qs"][i]))
    code["tmp_used"] = st["tmp_used"]
    prog["verifier_code"] = code

    # -- generate_fri_polynomial (starkinfo_fri_prover.rs:10-98), with the prover's ctx2ns
    vf1, vf2 = e_challenge("vf1"), e_challenge("vf2")
    fri = e_nop()
    for i in range(pil["nCommitments"]):
        fri = e_cm(i) if is_nop(fri) else e_add(e_mul(vf1, fri), e_cm(i))
    fri1, fri2 = e_nop(), e_nop()
    for i, ev in enumerate(info["ev_map"]):
        cur = fri2 if ev["prime"] else fri1
        e = {"cm": e_cm, "q": e_q, "const": e_const}[ev["type_"]](ev["id"])
        cur = e_sub(e, e_eval(i)) if is_nop(cur) else e_add(e_mul(cur, vf2), e_sub(e, e_eval(i)))
        if ev["prime"]:
            fri2 = cur
        else:
            fri1 = cur
    if not is_nop(fri):                                                   # sic: tests fri_exp (fri_prover.rs:64)
        fri1 = e_mul(fri1, E("xDivXSubXi"))
        fri = e_add(e_mul(vf1, fri), fri1) if not is_nop(fri) else fri1
    if not is_nop(fri2):
        fri2 = e_mul(fri2, E("xDivXSubWXi"))
        fri = e_add(e_mul(vf1, fri), fri2) if not is_nop(fri) else fri2
    info["fri_exp_id"] = len(pil["expressions"]); fri["keep2ns"] = True
    pil["expressions"].append(fri)
    pil_code_gen(ctx2ns, pil, info["fri_exp_id"], False, "f", 0, False)
    ctx2ns.code[-1]["code"][-1]["dest"] = Node("f", 0)
    prog["step52ns"] = build_code(ctx2ns, pil)

    # -- generate_fri_verifier (starkinfo_fri_ver.rs:7-21)
    ctx = Ctx()
    pil_code_gen(ctx, pil, info["fri_exp_id"], False, "", 0, True)
    prog["verifier_query_code"] = build_code(ctx, pil)
    info["n_exps"] = len(pil["expressions"])

    _map(info, pil, stark_struct, prog)
    info["publics"] = copy.deepcopy(pil["publics"])
    return info, prog, pil


# ---- StarkInfo::map (starkinfo_map.rs:10-307) -----------------------------------------------------
def _map(info, pil, stark_struct, prog):
    vpm = info["var_pol_map"]
    def add_pol(section, dim):
        vpm.append({"section": section, "section_pos": 0, "dim": dim, "exp_id": 0}); return len(vpm) - 1
    def add_cm(sec, dim):
        pn, p2 = add_pol(sec + "_n", dim), add_pol(sec + "_2ns", dim)
        info["cm_n"].append(pn); info["cm_2ns"].append(p2)
        info["map_sections"][sec + "_n"].append(pn); info["map_sections"][sec + "_2ns"].append(p2)
        return pn
    tmpexps = {}
    def im_none(i): return not info["im_exps"].get(i, False)
    def add_tmpexp(exp_id, dim):
        if im_none(exp_id) and exp_id not in tmpexps:
            tmpexps[exp_id] = len(info["tmpexp_n"])
            pp = add_pol("tmpexp_n", dim)
            info["tmpexp_n"].append(pp); info["map_sections"]["tmpexp_n"].append(pp); info["exp2pol"][exp_id] = pp
    n1, n2, n3, n4 = info["n_cm1"], info["n_cm2"], info["n_cm3"], info["n_cm4"]
    pil["cm_dims"] = [0] * (n1 + n2 + n3 + n4)
    for i in range(n1):
        add_cm("cm1", 1); pil["cm_dims"][i] = 1
    for i, pu in enumerate(info["pu_ctx"]):
        dim = max(exp_dim(pil, pil["expressions"][pu["f_exp_id"]]), exp_dim(pil, pil["expressions"][pu["t_exp_id"]]))
        add_cm("cm2", dim); pil["cm_dims"][n1 + i * 2] = dim
        add_cm("cm2", dim); pil["cm_dims"][n1 + i * 2 + 1] = dim
        add_tmpexp(pu["f_exp_id"], dim); add_tmpexp(pu["t_exp_id"], dim)
    for i, o in enumerate(info["pu_ctx"] + info["pe_ctx"] + info["ci_ctx"]):
        add_cm("cm3", 3); pil["cm_dims"][n1 + n2 + i] = 3
        add_tmpexp(o["num_id"], 3); add_tmpexp(o["den_id"], 3)
    for i, k in enumerate(info["im_exps_list"]):
        dim = exp_dim(pil, pil["expressions"][k])
        pn = add_cm("cm3", dim)
        pil["cm_dims"][n1 + n2 + i] = dim                                  # sic (starkinfo_map.rs:186)
        info["exp2pol"][k] = pn
    info["q_dim"] = exp_dim(pil, pil["expressions"][info["c_exp"]])
    for i in range(info["q_deg"]):
        add_cm("cm4", info["q_dim"]); pil["cm_dims"][n1 + n2 + n3 + i] = info["q_dim"]
    info["q_2ns"].append(add_pol("q_2ns", info["q_dim"]))
    info["f_2ns"].append(add_pol("f_2ns", 3))
    for s in SECTIONS:                                                     # map_section :490-516
        p = 0
        for e in (1, 2, 3):
            for pp in vpm:
                if pp["section"] == s and pp["dim"] == e:
                    pp["section_pos"] = p; p += e
            if e == 1:
                info["map_sectionsN1"][s] = p
            if e == 3:
                info["map_sectionsN"][s] = p
        info["map_sectionsN3"][s] = (info["map_sectionsN"][s] - info["map_sectionsN1"][s]) // 3
    N, Next = 1 << stark_struct["nBits"], 1 << stark_struct["nBitsExt"]
    off, sn = info["map_offsets"], info["map_sectionsN"]
    order = [("cm1_n", N), ("cm2_n", N), ("cm3_n", N), ("cm4_n", N), ("tmpexp_n", N), ("cm1_2ns", Next), ("cm2_2ns", Next),
             ("cm3_2ns", Next), ("cm4_2ns", Next), ("q_2ns", Next), ("f_2ns", Next)]
    acc = 0
    for s, deg in order:
        off[s] = acc; acc += deg * sn[s]; info["map_deg"][s] = deg
    info["map_total_n"] = acc

    def fix_prover_code(seg, dom):                                         # :427-488
        st = {"map": {}, "tmp_used": seg["tmp_used"]}
        def fix(r):
            t = r["type_"]
            if t == "cm":
                r["p"] = info["cm_n"][r["id"]] if dom == "n" else info["cm_2ns"][r["id"]]
            elif t == "exp":
                if r["id"] in info["im_exps_list"]:
                    r["type_"] = "cm"; r["id"] = info["im_exp2cm"][r["id"]]
                elif r["id"] in tmpexps and dom == "n":
                    r["type_"] = "tmpExp"; r["dim"] = exp_dim(pil, pil["expressions"][r["id"]]); r["id"] = tmpexps[r["id"]]
                else:
                    k = (1 if r["prime"] else 0, r["id"])
                    if k not in st["map"]:
                        st["map"][k] = st["tmp_used"]; st["tmp_used"] += 1
                    r["type_"] = "tmp"; r["exp_id"] = r["id"]; r["id"] = st["map"][k]
            elif t not in ("const", "number", "challenge", "public", "tmp", "Zi", "xDivXSubXi", "xDivXSubWXi", "eval", "x", "q", "f", "tmpExp"):
                raise ValueError("Invalid reference type " + t)
        iterate_code(seg, fix)
        seg["tmp_used"] = st["tmp_used"]
    for seg in prog["publics_code"]:
        fix_prover_code(seg, "n")
    for name, dom in (("step2prev", "n"), ("step3prev", "n"), ("step3", "n"), ("step42ns", "2ns"), ("step52ns", "2ns"),
                      ("verifier_query_code", "2ns")):
        fix_prover_code(prog[name], dom)
    def fix_tree(r):                                                       # :257-283
        if r["type_"] == "cm":
            p1 = vpm[info["cm_2ns"][r["id"]]]
            r["type_"] = {"cm1_2ns": "tree1", "cm2_2ns": "tree2", "cm3_2ns": "tree3", "cm4_2ns": "tree4"}[p1["section"]]
            r["tree_pos"] = p1["section_pos"]; r["dim"] = p1["dim"]
    iterate_code(prog["verifier_query_code"], fix_tree)

    def set_code_dimensions(seg, dim_x):                                   # :309-425
        tmp_dim = {}
        def get_dim(r):
            t = r["type_"]
            if t == "tmp": d = tmp_dim[r["id"]]
            elif t in ("tree1", "tree2", "tree3", "tree4", "tmpExp"): d = r["dim"]
            elif t == "cm": d = vpm[info["cm_2ns"][r["id"]]]["dim"]
            elif t == "q": d = vpm[info["qs"][r["id"]]]["dim"]
            elif t in ("const", "number", "public", "Zi"): d = 1
            elif t in ("eval", "challenge", "Z"): d = 3
            elif t in ("xDivXSubXi", "xDivXSubWXi", "x"): d = dim_x
            else: raise ValueError("Invalid reference type get " + t)
            if d == 0:
                raise ValueError("Invalid dim")
            r["dim"] = d
            return d
        for part in ("first", "i", "last"):
            for c in seg[part]:
                if c["op"] in ("add", "sub", "mul"): nd = max(get_dim(c["src"][0]), get_dim(c["src"][1]))
                elif c["op"] == "muladd": nd = max(get_dim(c["src"][0]), get_dim(c["src"][1]), get_dim(c["src"][2]))
                elif c["op"] == "copy": nd = get_dim(c["src"][0])
                else: raise ValueError("Invalid op: " + c["op"])
                d = c["dest"]
                if d["type_"] == "tmp":
                    tmp_dim[d["id"]] = nd; d["dim"] = nd
                elif d["type_"] in ("exp", "cm", "q", "tmpExp", "f"):
                    d["dim"] = nd
                else:
                    raise ValueError("Invalid reference type set " + d["type_"])
    for i in range(info["n_publics"]):
        if i < len(prog["publics_code"]):
            s = prog["publics_code"][i]
            if s["first"] or s["i"] or s["last"]:
                set_code_dimensions(s, 1)
    for name, dx in (("step2prev", 1), ("step3prev", 1), ("step3", 1), ("step42ns", 1), ("step52ns", 1),
                     ("verifier_code", 3), ("verifier_query_code", 1)):
        set_code_dimensions(prog[name], dx)


def to_json(info, prog):
    """serde_json shape: HashMap<usize,_> keys become strings, EVIdx maps become [[p,id],idx] lists."""
    out = copy.deepcopy(info)
    out["im_exps"] = {str(k): v for k, v in info["im_exps"].items()}
    out["im_exp2cm"] = {str(k): v for k, v in info["im_exp2cm"].items()}
    out["exp2pol"] = {str(k): v for k, v in info["exp2pol"].items()}
    out["ev_idx"] = {"cm": [[[p, i], v] for (p, i), v in info["ev_idx"]["cm"].items()],
                     "const_": [[[p, i], v] for (p, i), v in info["ev_idx"]["const_"].items()]}
    return {"starkinfo": out, "program": prog}
