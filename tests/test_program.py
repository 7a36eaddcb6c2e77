"""Constraint-evaluation programs: run-time compiled gfx950 kernels (zk_program_*) vs the Python
restatement of the reference interpreter (oracle/interp.py)."""
import pathlib
import sys
import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
P = 0xFFFFFFFF00000001

BUF = {"cm1": 0, "const": 1, "q": 2, "cm3": 3}


def _conv(zk, o):
    """oracle operand dict -> zk_operand"""
    k = o["kind"]
    if k == "tmp":
        return zk.opnd(zk.OPND_TMP, id=o["id"])
    if k == "mem":
        return zk.opnd(zk.OPND_MEM, id=o["id"], dim=o.get("dim", 1), prime=o.get("prime", False),
                       buf=BUF[o["buf"]], stride=o["stride"])
    if k == "number":
        return zk.opnd(zk.OPND_NUMBER, value=o["value"] % P)
    return zk.opnd({"public": zk.OPND_PUBLIC, "challenge": zk.OPND_CHALLENGE, "eval": zk.OPND_EVAL, "x": zk.OPND_X,
                    "Zi": zk.OPND_ZI, "xDivXSubXi": zk.OPND_XDIVXSUBXI, "xDivXSubWXi": zk.OPND_XDIVXSUBWXI}[k],
                   id=o.get("id", 0))


def _compile(zk, program):
    ops = {"add": zk.OP_ADD, "sub": zk.OP_SUB, "mul": zk.OP_MUL, "copy": zk.OP_COPY}
    return zk.Program([zk.instr(ops[op], _conv(zk, d), _conv(zk, a), _conv(zk, b) if b else None)
                       for op, d, a, b in program])


def T(i): return {"kind": "tmp", "id": i}
def M(buf, col, stride, dim=1, prime=False): return {"kind": "mem", "buf": buf, "id": col, "stride": stride, "dim": dim, "prime": prime}
def N(v): return {"kind": "number", "value": v}


def _fib_like_program():
    """Shape of a step-4 program: Fibonacci transition constraints combined with the challenge vc,
    multiplied by Zi, written to q (dim 3); plus a dim-1 and a dim-3 intermediate column."""
    ch = lambda i: {"kind": "challenge", "id": i}
    return [
        ("mul", T(0), M("cm1", 0, 2), M("cm1", 0, 2)),                 # l1^2
        ("mul", T(1), M("cm1", 1, 2), M("cm1", 1, 2)),                 # l2^2
        ("add", T(2), T(0), T(1)),
        ("sub", T(3), M("cm1", 0, 2, prime=True), T(2)),              # l1' - (l1^2 + l2^2)
        ("sub", T(4), M("cm1", 1, 2, prime=True), M("cm1", 0, 2)),    # l2' - l1
        ("sub", T(5), N(1), M("const", 0, 1)),                         # 1 - L_last
        ("mul", T(6), T(3), T(5)),
        ("mul", T(7), T(4), T(5)),
        ("mul", T(8), ch(4), T(6)),                                    # dim 3 * dim 1
        ("add", T(9), T(8), T(7)),                                     # dim 3 + dim 1
        ("mul", T(10), ch(4), T(9)),                                   # dim 3 * dim 3
        ("sub", T(11), {"kind": "public", "id": 0}, T(10)),            # dim 1 - dim 3
        ("add", T(12), T(11), {"kind": "x"}),
        ("copy", M("cm3", 0, 4), T(2), None),                          # dim-1 write
        ("copy", M("cm3", 1, 4, dim=3), T(9), None),                   # dim-3 write
        ("mul", T(13), M("cm3", 1, 4, dim=3), M("cm3", 0, 4)),         # read back own writes
        ("add", T(14), T(12), T(13)),
        ("mul", M("q", 0, 3, dim=3), T(14), {"kind": "Zi"}),
    ]


def test_program_compiles_without_gpu(zk):
    prog = _compile(zk, _fib_like_program())
    src = prog.source
    body = src[src.index("void zk_eval_kernel"):]
    assert "mul31" in body and "sub13" in body and body.count("= gl::mul(") == 4
    assert body.count("pacc(") == 4 and "gl::f3_mul" not in body       # vc * t6 + t7 and vc * (vc * t6 + t7): sums over powers of vc
    assert "void zk_pow_kernel" in src


def test_translator_lays_reads_first_and_stages_wide_sections(zk):
    """the kernel text (no GPU needed): every section read sits in the prologue, ahead of the arithmetic and of the stores;
    rows of 8 words and more come through stage_in, narrow ones and sparsely used ones are read directly; a program
    that reads more words than the prologue may hold keeps its reads where it uses them"""
    BUF["wide"] = 4
    rng = np.random.default_rng(3)
    src = _compile(zk, _wide_program(rng, 40, 9)).source
    body = src[src.index("void zk_eval_kernel"):]
    assert "stage_in<14, 40>" in body and "stage_in<12, 40>" in body                        # 40 columns: chunks of 14 + 14 + 12
    assert "stage_in<9, 9>" in body and "stage_in<16, 64>" not in body and "c.bufs[4][" in body
    first_store = body.index("if (live) {")
    assert body.rindex("stage_in<") < body.index("gl::") < first_store                    # reads, then arithmetic, then stores
    assert all(body.index(s) > first_store for s in ("c.bufs[3][i * 4ull + 0] =", "c.bufs[2][i * 3ull + 0] ="))
    narrow = _compile(zk, _fib_like_program()).source
    assert "stage_in<" not in narrow[narrow.index("void zk_eval_kernel"):]                # 2-column sections: direct reads
    big = _compile(zk, _wide_program(rng, 300, 20)).source
    assert "stage_in<" not in big[big.index("void zk_eval_kernel"):] and "zk_stage" not in big
    # a read that would need one of this lane's stores of another shape cannot be issued first: rejected
    with pytest.raises(zk.ZkError, match="partially overlaps an earlier write"):
        _compile(zk, [("copy", T(0), {"kind": "challenge", "id": 4}, None), ("copy", M("cm3", 0, 4, dim=3), T(0), None), ("copy", T(1), M("cm3", 1, 4), None)])


def test_program_rejects_bad_code(zk):
    with pytest.raises(zk.ZkError, match="tmp read before write"):
        _compile(zk, [("add", T(1), T(0), N(1))])
    with pytest.raises(zk.ZkError, match="written at one row and read at the next row"):
        _compile(zk, [("copy", M("cm3", 0, 4), N(1), None), ("copy", T(0), M("cm3", 0, 4, prime=True), None)])
    # a primed store followed by an unprimed read of the same column: lane i would read row i while lane i-next writes it
    with pytest.raises(zk.ZkError, match="written at one row and read at the next row"):
        _compile(zk, [("copy", M("cm3", 0, 4, prime=True), N(1), None), ("copy", T(0), M("cm3", 0, 4), None)])
    # overlapping cells of different ids: a dim-3 write at column 0 covers the cell a primed read takes at column 1
    with pytest.raises(zk.ZkError, match="written at one row and read at the next row"):
        _compile(zk, [("copy", T(0), {"kind": "challenge", "id": 4}, None), ("copy", M("cm3", 0, 4, dim=3), T(0), None), ("copy", T(1), M("cm3", 1, 4, prime=True), None)])
    # a primed store read back primed by the same lane is served from the lane's own value: accepted
    _compile(zk, [("copy", M("cm3", 0, 4, prime=True), N(1), None), ("copy", T(0), M("cm3", 0, 4, prime=True), None), ("copy", M("q", 0, 3), T(0), None)])


@pytest.mark.gpu
@pytest.mark.parametrize("nbits,ext", [(3, 1), (6, 1), (8, 2)])
def test_program_matches_reference_interpreter(zk, orc, nbits, ext):
    import interp
    assert zk.lib().zk_device_count() >= 1
    zk.init(0)
    rng = np.random.default_rng(nbits * 10 + ext)
    n = 1 << (nbits + ext); nxt = 1 << ext
    program = _fib_like_program()
    cm1 = rng.integers(0, P, size=2 * n, dtype=np.uint64)
    const = rng.integers(0, 2, size=n, dtype=np.uint64)
    chal = rng.integers(0, P, size=24, dtype=np.uint64)
    pub = rng.integers(0, P, size=2, dtype=np.uint64)
    zi = orc.zh_inv(nbits, ext)
    d = {"cm1": zk.DevArray.from_host(cm1), "const": zk.DevArray.from_host(const),
         "q": zk.DevArray(3 * n, zero=True), "cm3": zk.DevArray(4 * n, zero=True)}
    x = zk.x_table(nbits + ext, 49)
    _compile(zk, program).run({BUF[k]: v for k, v in d.items()}, nbits + ext, nxt, publics=zk.DevArray.from_host(pub),
                              challenges=zk.DevArray.from_host(chal), x=x, zi=zk.DevArray.from_host(zi))
    bufs = {"cm1": [int(v) for v in cm1], "const": [int(v) for v in const], "q": [0] * (3 * n), "cm3": [0] * (4 * n)}
    interp.run(program, bufs, n, nxt, publics=[int(v) for v in pub], challenges=chal.reshape(8, 3).astype(object).tolist(),
               x=[int(v) for v in x.to_host()], zi=[int(v) for v in zi])
    assert [int(v) for v in d["q"].to_host()] == bufs["q"]
    assert [int(v) for v in d["cm3"].to_host()] == bufs["cm3"]


@pytest.mark.gpu
def test_row_range_run_touches_only_its_rows(zk, orc):
    """zk_program_run_rows_dev: rows row0..row0+count-1 get what the whole-domain run gives them (primed reads still wrap
    around the whole domain), every other destination cell is left alone -- how a public calculator is evaluated at its
    one row (stark_gen.rs:558-572)"""
    assert zk.lib().zk_device_count() >= 1
    zk.init(0)
    nbits, ext = 6, 1
    rng = np.random.default_rng(99)
    n = 1 << (nbits + ext); nxt = 1 << ext
    program = _fib_like_program()
    cm1 = rng.integers(0, P, size=2 * n, dtype=np.uint64)
    const = rng.integers(0, 2, size=n, dtype=np.uint64)
    chal = zk.DevArray.from_host(rng.integers(0, P, size=24, dtype=np.uint64))
    pub = zk.DevArray.from_host(rng.integers(0, P, size=2, dtype=np.uint64))
    zi = zk.DevArray.from_host(orc.zh_inv(nbits, ext)); x = zk.x_table(nbits + ext, 49)
    prog = _compile(zk, program)
    def go(rows):
        d = {"cm1": zk.DevArray.from_host(cm1), "const": zk.DevArray.from_host(const),
             "q": zk.DevArray.from_host(np.full(3 * n, 7, dtype=np.uint64)), "cm3": zk.DevArray.from_host(np.full(4 * n, 7, dtype=np.uint64))}
        prog.run({BUF[k]: v for k, v in d.items()}, nbits + ext, nxt, publics=pub, challenges=chal, x=x, zi=zi, rows=rows)
        return d["q"].to_host().reshape(n, 3), d["cm3"].to_host().reshape(n, 4)
    q_all, c_all = go(None)
    for row0, count in [(0, 1), (n - 1, 1), (n - 3, 3), (17, 40), (0, n)]:
        q, c = go((row0, count))
        inside = np.zeros(n, dtype=bool); inside[row0:row0 + count] = True
        assert (q[inside] == q_all[inside]).all() and (q[~inside] == 7).all()
        assert (c[inside] == c_all[inside]).all() and (c[~inside] == 7).all()
    with pytest.raises(zk.ZkError, match="outside the domain"):
        go((n - 1, 2))


def _long_chain_program(n_terms):
    """acc <- v * acc + cm1[j] over more terms than one group of the deferred sum holds, then (column - eval) terms on
    a second challenge, as the FRI polynomial's generated code does"""
    ch = lambda i: {"kind": "challenge", "id": i}
    ev = lambda i: {"kind": "eval", "id": i}
    prog = [("mul", T(0), ch(2), M("cm1", 0, 2))]
    t = 0
    for j in range(1, n_terms):
        prog.append(("add", T(t + 1), T(t), M("cm1", j % 2, 2, prime=bool(j % 3 == 0))))
        prog.append(("mul", T(t + 2), T(t + 1), ch(2)))
        t += 2
    prog.append(("sub", T(t + 1), M("cm1", 0, 2), ev(0)))
    prog.append(("mul", T(t + 2), T(t + 1), ch(5)))
    prog.append(("sub", T(t + 3), M("cm1", 1, 2), ev(3)))
    prog.append(("add", T(t + 4), T(t + 2), T(t + 3)))
    prog.append(("mul", T(t + 5), ch(5), T(t + 4)))
    prog.append(("add", T(t + 6), T(t), T(t + 5)))
    prog.append(("mul", M("q", 0, 3, dim=3), T(t + 6), {"kind": "xDivXSubXi"}))
    return prog


@pytest.mark.gpu
@pytest.mark.parametrize("n_terms", [3, 700, 1300])
def test_long_horner_chains_match_reference_interpreter(zk, orc, n_terms):
    import interp
    zk.init(0)
    nbits, ext = 4, 1
    rng = np.random.default_rng(n_terms)
    n = 1 << (nbits + ext); nxt = 1 << ext
    program = _long_chain_program(n_terms)
    cm1 = rng.integers(0, P, size=2 * n, dtype=np.uint64)
    cm1[:4] = [0, P - 1, 1, P - 2]
    chal = rng.integers(0, P, size=24, dtype=np.uint64); evals = rng.integers(0, P, size=12, dtype=np.uint64)
    xd = rng.integers(0, P, size=3 * n, dtype=np.uint64)
    d = {"cm1": zk.DevArray.from_host(cm1), "q": zk.DevArray(3 * n, zero=True)}
    prog = _compile(zk, program)
    assert prog.source.count("pfin(") >= 1 + (n_terms + 511) // 512
    prog.run({BUF[k]: v for k, v in d.items()}, nbits + ext, nxt, challenges=zk.DevArray.from_host(chal),
             evals=zk.DevArray.from_host(evals), xdiv=zk.DevArray.from_host(xd))
    bufs = {"cm1": [int(v) for v in cm1], "q": [0] * (3 * n)}
    interp.run(program, bufs, n, nxt, challenges=chal.reshape(8, 3).astype(object).tolist(), evals=evals.reshape(4, 3).astype(object).tolist(),
               xdiv=[int(v) for v in xd])
    assert [int(v) for v in d["q"].to_host()] == bufs["q"]


def _random_program(rng, n_ops):
    """A random straight-line step program over every operand kind the prover's segments use (A.9): temporaries of both
    dimensions, primed and unprimed column reads, numbers, publics, challenges, evals, x, Zi, x/(x - xi) tables, Horner
    chains on a challenge (the shape the translator keeps symbolic), read-back of its own column writes; it ends by
    writing a dim-1 and a dim-3 column of cm3 and the dim-3 q."""
    ch = lambda i: {"kind": "challenge", "id": i}
    leaves1 = [lambda: M("cm1", int(rng.integers(0, 4)), 4, prime=bool(rng.integers(0, 2))),
               lambda: M("const", int(rng.integers(0, 2)), 2, prime=bool(rng.integers(0, 2))),
               lambda: N([0, 1, 2, 7, P - 1, 1 << 40, int(rng.integers(0, P, dtype=np.uint64))][int(rng.integers(0, 7))]),
               lambda: {"kind": "public", "id": int(rng.integers(0, 2))},
               lambda: {"kind": "x"}, lambda: {"kind": "Zi"}]
    leaves3 = [lambda: ch(int(rng.integers(0, 8))), lambda: {"kind": "eval", "id": int(rng.integers(0, 4))},
               lambda: M("cm1", 1, 4, dim=3, prime=bool(rng.integers(0, 2))),
               lambda: {"kind": "xDivXSubXi"}, lambda: {"kind": "xDivXSubWXi"}]
    prog, dims = [], []                                                # dims[i] = dimension of T(i)

    def operand():
        r = rng.random()
        if dims and r < 0.5:
            i = int(rng.integers(0, len(dims))); return T(i), dims[i]
        if r < 0.85:
            return leaves1[int(rng.integers(0, len(leaves1)))](), 1
        return leaves3[int(rng.integers(0, len(leaves3)))](), 3

    def emit(op, a, b):
        d = max(a[1], b[1]) if b else a[1]
        prog.append((op, T(len(dims)), a[0], b[0] if b else None)); dims.append(d)
        return T(len(dims) - 1), d

    while len(prog) < n_ops:
        r = rng.random()
        if r < 0.2:                                                      # acc <- v * acc + d, d a column, a (column - eval) or a temporary
            v = ch(int(rng.integers(0, 8)))
            acc = emit("mul", (v, 3), operand())
            for _ in range(int(rng.integers(2, 40))):
                if rng.random() < 0.3:
                    d = emit("sub", (M("cm1", int(rng.integers(0, 4)), 4, prime=bool(rng.integers(0, 2))), 1), ({"kind": "eval", "id": int(rng.integers(0, 4))}, 3))
                else:
                    d = operand()
                acc = emit("add", acc, d) if rng.random() < 0.8 else emit("sub", acc, d)
                acc = emit("mul", (v, 3), acc) if rng.random() < 0.5 else emit("mul", acc, (v, 3))
        elif r < 0.3:
            emit("copy", operand(), None)
        else:
            emit(str(rng.choice(["add", "sub", "mul"])), operand(), operand())
    one = [i for i, d in enumerate(dims) if d == 1]; three = [i for i, d in enumerate(dims) if d == 3]
    if not one: emit("add", (leaves1[0](), 1), (N(3), 1)); one = [len(dims) - 1]
    if not three: emit("mul", (ch(0), 3), (T(one[-1]), 1)); three = [len(dims) - 1]
    prog.append(("copy", M("cm3", 0, 4), T(one[int(rng.integers(0, len(one)))]), None))
    prog.append(("copy", M("cm3", 1, 4, dim=3), T(three[int(rng.integers(0, len(three)))]), None))
    back = emit("mul", (M("cm3", 1, 4, dim=3), 3), (M("cm3", 0, 4), 1))                                  # read back own writes
    last = emit("add", back, (T(three[-1]), 3))
    prog.append(("mul", M("q", 0, 3, dim=3), last[0], {"kind": "Zi"}))
    return prog


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_random_programs_match_reference_interpreter(zk, orc, seed):
    import interp
    zk.init(0)
    rng = np.random.default_rng(1000 + seed)
    nbits, ext = 4, 1
    n = 1 << (nbits + ext); nxt = 1 << ext
    program = _random_program(rng, int(rng.integers(5, 120)))
    cm1 = rng.integers(0, P, size=4 * n, dtype=np.uint64); cm1[:6] = [0, P - 1, 1, P - 2, 0, 0]
    const = rng.integers(0, P, size=2 * n, dtype=np.uint64)
    chal = rng.integers(0, P, size=24, dtype=np.uint64); evals = rng.integers(0, P, size=12, dtype=np.uint64)
    if seed % 3 == 0: chal[3:6] = [5, 0, 0]                                # a base-field valued challenge
    pub = rng.integers(0, P, size=2, dtype=np.uint64)
    xd, xdw = rng.integers(0, P, size=3 * n, dtype=np.uint64), rng.integers(0, P, size=3 * n, dtype=np.uint64)
    zi = orc.zh_inv(nbits, ext); x = zk.x_table(nbits + ext, 49)
    d = {"cm1": zk.DevArray.from_host(cm1), "const": zk.DevArray.from_host(const), "q": zk.DevArray(3 * n, zero=True), "cm3": zk.DevArray(4 * n, zero=True)}
    _compile(zk, program).run({BUF[k]: v for k, v in d.items()}, nbits + ext, nxt, publics=zk.DevArray.from_host(pub), challenges=zk.DevArray.from_host(chal),
                              evals=zk.DevArray.from_host(evals), x=x, zi=zk.DevArray.from_host(zi), xdiv=zk.DevArray.from_host(xd), xdivw=zk.DevArray.from_host(xdw))
    bufs = {"cm1": [int(v) for v in cm1], "const": [int(v) for v in const], "q": [0] * (3 * n), "cm3": [0] * (4 * n)}
    interp.run(program, bufs, n, nxt, publics=[int(v) for v in pub], challenges=chal.reshape(8, 3).astype(object).tolist(),
               evals=evals.reshape(4, 3).astype(object).tolist(), x=[int(v) for v in x.to_host()], zi=[int(v) for v in zi],
               xdiv=[int(v) for v in xd], xdivw=[int(v) for v in xdw])
    assert [int(v) for v in d["cm3"].to_host()] == bufs["cm3"]
    assert [int(v) for v in d["q"].to_host()] == bufs["q"]


def _wide_program(rng, w_cm1, w_const):
    """Reads nearly every column of two wide sections -- base-field cells and cubic-extension cells that straddle the
    chunks the translator stages through LDS, at row i and at row i+next -- and a few columns of a third one (too few
    for staging); writes a dim-1 and a dim-3 column of cm3 and q."""
    prog, t = [], 0
    prog.append(("mul", T(0), M("cm1", 0, w_cm1), M("const", 0, w_const, prime=True)))
    for col in range(1, w_cm1):
        src = M("cm1", col, w_cm1, prime=bool(rng.integers(0, 2)))
        prog.append((str(rng.choice(["add", "sub", "mul"])), T(t + 1), T(t), src)); t += 1
    for col in range(1, w_const):
        prog.append((str(rng.choice(["add", "sub"])), T(t + 1), T(t), M("const", col, w_const, prime=bool(rng.integers(0, 2))))); t += 1
    d1 = t
    for col in sorted({1, 12, 13, 14, 17, 18, 19, w_cm1 - 3}):                # cubic-extension cells across chunk boundaries
        if col + 3 > w_cm1: continue
        prog.append(("mul", T(t + 1), M("cm1", col, w_cm1, dim=3, prime=bool(rng.integers(0, 2))), T(t))); t += 1
    prog.append(("add", T(t + 1), T(t), M("wide", 5, 64)));  t += 1            # 2 of 64 columns: read directly
    prog.append(("add", T(t + 1), T(t), M("wide", 40, 64, dim=3, prime=True)));  t += 1
    prog.append(("copy", M("cm3", 0, 4), T(d1), None))
    prog.append(("copy", M("cm3", 1, 4, dim=3), T(t), None))
    prog.append(("mul", M("q", 0, 3, dim=3), T(t), {"kind": "x"}))
    return prog


@pytest.mark.gpu
@pytest.mark.parametrize("w_cm1,w_const,nbits", [(19, 18, 9), (40, 9, 8), (36, 8, 7), (73, 20, 8), (300, 20, 7)])
def test_wide_sections_match_reference_interpreter(zk, orc, w_cm1, w_const, nbits):
    """Sections of 8 and more columns are read through LDS in chunks of at most 19 columns (csrc/expr_jit.hip stage_in):
    whole domain (several waves and blocks, the primed rows of the last wave wrapping to row 0) and a row range that
    starts and ends inside a wave.  A program that reads more words than the kernel can hold in registers (the last
    case) reads them where it uses them instead."""
    import interp
    assert zk.lib().zk_device_count() >= 1
    zk.init(0)
    BUF["wide"] = 4
    rng = np.random.default_rng(w_cm1 * 100 + w_const)
    ext = 1
    n = 1 << (nbits + ext); nxt = 1 << ext
    program = _wide_program(rng, w_cm1, w_const)
    host = {"cm1": rng.integers(0, P, size=w_cm1 * n, dtype=np.uint64), "const": rng.integers(0, P, size=w_const * n, dtype=np.uint64),
            "wide": rng.integers(0, P, size=64 * n, dtype=np.uint64)}
    chal = rng.integers(0, P, size=24, dtype=np.uint64)
    x = zk.x_table(nbits + ext, 49); zi = orc.zh_inv(nbits, ext)
    prog = _compile(zk, program)
    words = {(o["buf"], o["id"] + j, bool(o.get("prime"))) for _, _, a, b in program for o in (a, b)
             if o is not None and o["kind"] == "mem" for j in range(o.get("dim", 1))}       # distinct (section, word, row i / i + next) reads
    if len(words) <= 112:                                                  # Gen::HOIST_MAX: every word read once, up front, wide sections through LDS
        assert "stage_in<" in prog.source and "c.bufs[4][" in prog.source
    else:
        assert "stage_in<" not in prog.source
    def go(rows):
        d = {k: zk.DevArray.from_host(v) for k, v in host.items()}
        d["q"] = zk.DevArray.from_host(np.full(3 * n, 7, dtype=np.uint64)); d["cm3"] = zk.DevArray.from_host(np.full(4 * n, 7, dtype=np.uint64))
        prog.run({BUF[k]: v for k, v in d.items()}, nbits + ext, nxt, challenges=zk.DevArray.from_host(chal), x=x, zi=zk.DevArray.from_host(zi), rows=rows)
        return d["q"].to_host().reshape(n, 3), d["cm3"].to_host().reshape(n, 4)
    bufs = {k: [int(v) for v in a] for k, a in host.items()}
    bufs["q"] = [0] * (3 * n); bufs["cm3"] = [0] * (4 * n)
    interp.run(program, bufs, n, nxt, challenges=chal.reshape(8, 3).astype(object).tolist(), x=[int(v) for v in x.to_host()], zi=[int(v) for v in zi])
    q_ref = np.array(bufs["q"], dtype=np.uint64).reshape(n, 3); c_ref = np.array(bufs["cm3"], dtype=np.uint64).reshape(n, 4)
    q, c = go(None)
    assert (q == q_ref).all() and (c == c_ref).all()
    for row0, count in [(n - 70, 70), (37, 300 if n > 400 else 100), (5, 1)]:
        q, c = go((row0, count))
        inside = np.zeros(n, dtype=bool); inside[row0:row0 + count] = True
        assert (q[inside] == q_ref[inside]).all() and (q[~inside] == 7).all()
        assert (c[inside] == c_ref[inside]).all() and (c[~inside] == 7).all()


_JIT_PROBE = r'''
import sys, pathlib
import numpy as np
ROOT = pathlib.Path(sys.argv[1]); sys.path.insert(0, str(ROOT / "tests"))
import zkgpu_loader, test_program as TP
zk = zkgpu_loader.load()
def stats():
    o = np.zeros(3, np.uint64); zk.lib().zk_jit_cache_stats(o.ctypes.data); return [int(v) for v in o]
p = TP._compile(zk, TP._fib_like_program()); a = stats()
q = TP._compile(zk, TP._fib_like_program()); b = stats()          # same text: the in-process cache
assert p.source == q.source
print("first", *a, "second", *b)
'''


def test_code_objects_are_cached_in_the_process_and_on_disk(tmp_path):
    """csrc/expr_jit.hip compile_cached: a step program is compiled by hipRTC once per text -- the second compilation in a process
    comes from memory, the first one of the next process from $ZK_JIT_CACHE/<sha256>.co; a damaged file is recompiled and replaced;
    ZK_JIT_CACHE=off writes nothing.  (Compiling needs no GPU.)  Counted by zk_jit_cache_stats: {hipRTC, disk, memory}."""
    import hashlib, os, subprocess, sys
    cache = tmp_path / "jit"
    def run(env_cache):
        env = dict(os.environ, ZK_JIT_CACHE=str(env_cache))
        r = subprocess.run([sys.executable, "-c", _JIT_PROBE, str(ROOT)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        w = r.stdout.split()
        return [int(x) for x in w[1:4]], [int(x) for x in w[5:8]]
    assert run(cache) == ([1, 0, 0], [1, 0, 1])                         # compiled once, then from memory
    files = sorted(cache.glob("*.co"))
    good = files[0].read_bytes()
    assert len(files) == 1 and len(files[0].stem) == 64 and good[:8] == b"ZKCO0001" and good[48:52] == b"\x7fELF"
    assert int.from_bytes(good[8:16], "little") == len(good) - 48 and hashlib.sha256(good[48:]).digest() == good[16:48]
    assert run(cache) == ([0, 1, 0], [0, 1, 1])                         # a fresh process: from disk
    for damaged in (b"not a code object",                               # foreign file
                    good[:len(good) // 2],                              # truncated: the ELF magic is there, the length and digest are not
                    good[:100] + bytes([good[100] ^ 1]) + good[101:],    # one flipped bit in the code object
                    good[48:]):                                         # a bare code object without the header (e.g. planted)
        files[0].write_bytes(damaged)                                   # compiled again, file replaced
        assert run(cache) == ([1, 0, 0], [1, 0, 1]) and files[0].read_bytes() == good
    assert run("off") == ([1, 0, 0], [1, 0, 1]) and len(list(cache.glob("*"))) == 1
    os.chmod(cache, 0o777)                                             # a directory others may write to is not trusted: nothing read, nothing written
    files[0].unlink()
    assert run(cache) == ([1, 0, 0], [1, 0, 1]) and list(cache.glob("*")) == []
    os.chmod(cache, 0o700)


def test_step_programs_can_compile_in_helper_processes(tmp_path):
    """csrc/expr_jit.hip compile_spawned: hipRTC compiles serially inside one process, so a setup hands each step program to a
    `zkgpu_jitc` process (csrc/jitc_main.cpp).  ZK_JIT_SPAWN=1 sends every compilation that way: the helper is really started (a
    wrapper named by $ZK_JITC leaves a mark), the code object it returns is the one the in-process compiler makes (same cache file,
    byte for byte), and a helper that cannot be started falls back to the in-process compiler.  (No GPU needed.)"""
    import os, stat, subprocess, sys
    real = ROOT / "eigen-zkvm_amd" / "zkgpu_jitc"
    assert real.exists(), "make -C eigen-zkvm_amd/csrc builds it"
    mark = tmp_path / "mark"
    wrapper = tmp_path / "jitc.sh"
    wrapper.write_text("#!/bin/sh\necho run >> %s\nexec %s \"$@\"\n" % (mark, real))
    wrapper.chmod(wrapper.stat().st_mode | stat.S_IXUSR)
    def run(cache, **env):
        e = dict(os.environ, ZK_JIT_CACHE=str(cache), **env)
        r = subprocess.run([sys.executable, "-c", _JIT_PROBE, str(ROOT)], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        w = r.stdout.split()
        return [int(x) for x in w[1:4]], [int(x) for x in w[5:8]]
    a, b = tmp_path / "a", tmp_path / "b"
    assert run(a, ZK_JIT_SPAWN="1", ZK_JITC=str(wrapper)) == ([1, 0, 0], [1, 0, 1]) and mark.read_text() == "run\n"
    assert run(b) == ([1, 0, 0], [1, 0, 1])                              # in process
    fa, fb = sorted(a.glob("*.co")), sorted(b.glob("*.co"))
    assert len(fa) == 1 and len(fb) == 1 and fa[0].name == fb[0].name and fa[0].read_bytes() == fb[0].read_bytes()
    assert run(tmp_path / "c", ZK_JIT_SPAWN="1", ZK_JITC=str(tmp_path / "missing")) == ([1, 0, 0], [1, 0, 1])   # no helper: in process
    assert mark.read_text() == "run\n"
