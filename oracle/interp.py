"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Pure-Python restatement of the reference's constraint
interpreter for small domains: starky/src/interpreter.rs:91-175 (Block::eval), :228-234 (get_i),
:236-283 (get_value) and the F3G runtime-dim arithmetic of starky/src/f3g.rs:323-449.

A value is a tuple of 1 or 3 ints (the reference's F3G with dim 1 or 3).  A program is a list of
(op, dest, src0, src1) with operands given as dicts:
  {"kind": "tmp", "id": k}
  {"kind": "mem", "buf": name, "id": column, "stride": width, "dim": 1|3, "prime": bool}
  {"kind": "number", "value": v} | {"kind": "public"|"challenge"|"eval", "id": k}
  {"kind": "x"} | {"kind": "Zi"} | {"kind": "xDivXSubXi"} | {"kind": "xDivXSubWXi"}
Rows are evaluated sequentially, as the reference does inside a chunk (stark_gen.rs:752-783).
"""
P = 0xFFFFFFFF00000001


def f3_mul(a, b):  # f3g.rs:420-430
    A = (a[0] + a[1]) * (b[0] + b[1]) % P
    B = (a[0] + a[2]) * (b[0] + b[2]) % P
    C = (a[1] + a[2]) * (b[1] + b[2]) % P
    D, E, F = a[0] * b[0] % P, a[1] * b[1] % P, a[2] * b[2] % P
    G = (D - E) % P
    return ((C + G - F) % P, (A + C - E - E - D) % P, (B - G) % P)


def v_add(a, b):  # f3g.rs:323-361
    if len(a) == 3 and len(b) == 3:
        return tuple((x + y) % P for x, y in zip(a, b))
    if len(a) == 3:
        return ((a[0] + b[0]) % P, a[1], a[2])
    if len(b) == 3:
        return ((b[0] + a[0]) % P, b[1], b[2])
    return ((a[0] + b[0]) % P,)


def v_sub(a, b):  # f3g.rs:370-398
    if len(a) == 3 and len(b) == 3:
        return tuple((x - y) % P for x, y in zip(a, b))
    if len(a) == 3:
        return ((a[0] - b[0]) % P, a[1], a[2])
    if len(b) == 3:
        return ((a[0] - b[0]) % P, (-b[1]) % P, (-b[2]) % P)
    return ((a[0] - b[0]) % P,)


def v_mul(a, b):  # f3g.rs:407-449
    if len(a) == 3 and len(b) == 3:
        return f3_mul(a, b)
    if len(a) == 3:
        return tuple(x * b[0] % P for x in a)
    if len(b) == 3:
        return tuple(x * a[0] % P for x in b)
    return (a[0] * b[0] % P,)


def run(program, bufs, n, next_, publics=(), challenges=(), evals=(), x=None, zi=None, xdiv=None, xdivw=None):
    """bufs: dict name -> flat list of ints (mutated in place)."""
    def get(o, i, tmp):
        k = o["kind"]
        if k == "tmp":
            return tmp[o["id"]]
        if k == "mem":
            row = (i + (next_ if o.get("prime") else 0)) % n                     # interpreter.rs:228-234
            base = o["id"] + row * o["stride"]
            b = bufs[o["buf"]]
            return (b[base],) if o.get("dim", 1) == 1 else (b[base], b[base + 1], b[base + 2])
        if k == "number":
            return (o["value"] % P,)
        if k == "public":
            return (publics[o["id"]],)
        if k == "challenge":
            return tuple(challenges[o["id"]])
        if k == "eval":
            return tuple(evals[o["id"]])
        if k == "x":
            return (x[i],)
        if k == "Zi":
            return (zi[i % len(zi)],)
        if k == "xDivXSubXi":
            return tuple(xdiv[3 * i:3 * i + 3])
        if k == "xDivXSubWXi":
            return tuple(xdivw[3 * i:3 * i + 3])
        raise ValueError(k)

    for i in range(n):
        tmp = {}
        for op, dest, s0, s1 in program:
            a = get(s0, i, tmp)
            if op == "copy":
                r = a
            else:
                b = get(s1, i, tmp)
                r = {"add": v_add, "sub": v_sub, "mul": v_mul}[op](a, b)
            if dest["kind"] == "tmp":
                tmp[dest["id"]] = r
            else:                                                                   # interpreter.rs:143-166
                row = (i + (next_ if dest.get("prime") else 0)) % n                 # get_i on the destination too
                base = dest["id"] + row * dest["stride"]
                buf = bufs[dest["buf"]]
                for j, v in enumerate(r):
                    buf[base + j] = v


def run_at(program, bufs, n, next_, i, **kw):
    """Evaluate the program at the single row i and return its last destination value
    (compile_code(..., ret = true) + Block::eval, interpreter.rs:218-223; stark_gen.rs:558-572)."""
    last = {}
    prog2 = list(program)
    # capture: run on a shadow that records the final destination
    res = {}
    def rec_run():
        tmp_bufs = bufs
        # re-implement the row loop for one row, recording the last written value
        def get(o, tmp):
            k = o["kind"]
            if k == "tmp":
                return tmp[o["id"]]
            if k == "mem":
                row = (i + (next_ if o.get("prime") else 0)) % n
                base = o["id"] + row * o["stride"]
                b = tmp_bufs[o["buf"]]
                return (b[base],) if o.get("dim", 1) == 1 else (b[base], b[base + 1], b[base + 2])
            if k == "number":
                return (o["value"] % P,)
            if k == "public":
                return (kw["publics"][o["id"]],)
            if k == "challenge":
                return tuple(kw["challenges"][o["id"]])
            if k == "x":
                return (kw["x"][i],)
            raise ValueError(k)
        tmp = {}
        r = None
        for op, dest, s0, s1 in prog2:
            a = get(s0, tmp)
            r = a if op == "copy" else {"add": v_add, "sub": v_sub, "mul": v_mul}[op](a, get(s1, tmp))
            if dest["kind"] == "tmp":
                tmp[dest["id"]] = r
            else:
                raise ValueError("public calculator writes to a section")
        return r
    return rec_run()


# ---- the same programs through oracle/interp.c (sizes the Python loop cannot reach) -----------------------------
_KIND = {"tmp": 0, "mem": 1, "number": 2, "public": 3, "challenge": 4, "eval": 5, "x": 6, "Zi": 7, "xDivXSubXi": 8, "xDivXSubWXi": 9}
_OP = {"add": 0, "sub": 1, "mul": 2, "copy": 3}


def encode(program, buf_index):
    """[(op, dest, src0, src1)] -> flat int64 words for orc_interp_run (19 per instruction); returns (words, n_tmp)"""
    import numpy as np

    def enc(o):
        if o is None:
            return [2, 0, 0, 0, 0, 0]
        k = o["kind"]
        if k == "tmp":
            return [0, o["id"], 0, 0, 0, 0]
        if k == "mem":
            return [1, buf_index[o["buf"]], o["id"], o["stride"], o.get("dim", 1), 1 if o.get("prime") else 0]
        if k == "number":
            v = o["value"] % P
            return [2, v - (1 << 64) if v >= (1 << 63) else v, 0, 0, 0, 0]
        if k in ("public", "challenge", "eval"):
            return [_KIND[k], o["id"], 0, 0, 0, 0]
        return [_KIND[k], 0, 0, 0, 0, 0]
    words, n_tmp = [], 0
    for op, dest, s0, s1 in program:
        words += [_OP[op]] + enc(dest) + enc(s0) + enc(s1)
        for o in (dest, s0, s1):
            if o is not None and o["kind"] == "tmp":
                n_tmp = max(n_tmp, o["id"] + 1)
    return np.array(words, np.int64), n_tmp


def run_c(lib, program, bufs, n, next_, publics=(), challenges=(), evals=(), x=None, zi=None, xdiv=None, xdivw=None):
    """bufs: dict name -> numpy uint64 array (mutated in place by the C interpreter)"""
    import ctypes as C
    import numpy as np
    names = sorted(bufs)
    code, n_tmp = encode(program, {k: i for i, k in enumerate(names)})
    arr = lambda v: np.ascontiguousarray(np.asarray(v if v is not None and len(v) else [0], dtype=np.uint64).reshape(-1))
    ptrs = (C.c_void_p * max(1, len(names)))(*[bufs[k].ctypes.data for k in names])
    for k in names:
        assert bufs[k].dtype == np.uint64 and bufs[k].flags["C_CONTIGUOUS"]
    pub, ch, ev, xx, zz, xd, xw = arr(publics), arr(challenges), arr(evals), arr(x), arr(zi), arr(xdiv), arr(xdivw)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.orc_interp_run.restype = C.c_int
    rc = lib.orc_interp_run(p(code), C.c_uint64(len(program)), C.c_uint64(n_tmp), ptrs, C.c_uint64(len(names)), C.c_uint64(n), C.c_uint64(next_),
                            p(pub), p(ch), p(ev), p(xx), p(zz), C.c_uint64(0 if zi is None else len(zz)), p(xd), p(xw))
    if rc:
        raise ValueError("orc_interp_run: malformed program")
