"""Synthetic workload generator: a "wide Fibonacci" PIL (compiled-PIL JSON, the shape pilcom emits,
cf. starky/data/fib.pil.json.gl) with W independent Fibonacci pairs -> 2W committed columns, one
constant column (ISLAST), 2W+1 identities of degree 2, one public.  W = 10 approximates the column
count of BASELINE's Poseidon PIL (19 committed columns) for which no PIL compiler is available here.
Input generation only -- nothing here is measured or shipped."""


def _num(v): return {"op": "number", "deg": 0, "value": str(v)}
def _cm(i, nxt=False): return {"op": "cm", "deg": 1, "id": i, "next": nxt}
def _const(i): return {"op": "const", "deg": 1, "id": i, "next": False}
def _op(op, deg, a, b): return {"op": op, "deg": deg, "values": [a, b]}


def wide_fib_pil(nbits, W):
    N = 1 << nbits
    refs = {"Fibonacci.ISLAST": {"type": "constP", "id": 0, "polDeg": N, "isArray": False}}
    exprs, ids = [], []
    not_last = lambda: _op("sub", 1, _num(1), _const(0))
    for k in range(W):
        a, b = 2 * k, 2 * k + 1
        refs["Fibonacci.a%d" % k] = {"type": "cmP", "id": a, "polDeg": N, "isArray": False}
        refs["Fibonacci.b%d" % k] = {"type": "cmP", "id": b, "polDeg": N, "isArray": False}
        exprs.append(_op("sub", 2, _op("mul", 2, not_last(), _op("sub", 1, _cm(a, True), _cm(b))), _num(0)))
        exprs.append(_op("sub", 2, _op("mul", 2, not_last(), _op("sub", 1, _cm(b, True), _op("add", 1, _cm(a), _cm(b)))), _num(0)))
    exprs.append(_op("sub", 2, _op("mul", 2, _const(0), _op("sub", 1, _cm(1), {"op": "public", "deg": 0, "id": 0})), _num(0)))
    return {"nCommitments": 2 * W, "nQ": 0, "nIm": 0, "nConstants": 1,
            "publics": [{"polType": "cmP", "polId": 1, "idx": N - 1, "id": 0, "name": "out"}],
            "references": refs, "expressions": exprs,
            "polIdentities": [{"e": i, "fileName": "widefib.pil", "line": i + 1} for i in range(len(exprs))],
            "plookupIdentities": [], "permutationIdentities": [], "connectionIdentities": []}


def stark_struct(nbits, n_queries=8):
    """steps like the reference's r2.starkStruct (fold by at most 5 bits per step, last step 2^4..2^5)"""
    ext = nbits + 1
    steps, b = [ext], ext
    while b > 5:
        b = max(b - 5, 4) if b - 5 >= 4 else 4
        steps.append(b)
    return {"nBits": nbits, "nBitsExt": ext, "nQueries": n_queries, "verificationHashType": "GL",
            "steps": [{"nBits": s} for s in steps]}


def const_trace(nbits):
    import numpy as np
    c = np.zeros(1 << nbits, np.uint64); c[-1] = 1
    return c


def program(nbits, W=10, hash_type="GL"):
    """({"starkinfo", "program"}, stark_struct) of the wide-Fibonacci PIL from the product's code generator"""
    import poseidong
    ss = stark_struct(nbits); ss["verificationHashType"] = hash_type
    return poseidong.native_program(wide_fib_pil(nbits, W), ss), ss


def wide_fib_trace(nbits, W, seed=0):
    import ctypes, pathlib
    import numpy as np
    lib = ctypes.CDLL(str(pathlib.Path(__file__).resolve().parent / "libtracegen.so"))
    out = np.zeros((1 << nbits) * 2 * W, np.uint64)
    lib.widefib_trace_seed(ctypes.c_uint(nbits), ctypes.c_uint(W), ctypes.c_uint64(seed), out.ctypes.data_as(ctypes.c_void_p))
    return out
