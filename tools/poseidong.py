"""Inputs of BASELINE config 3 / the 2^24 headline: the PoseidonG state machine (`starkjs/poseidon/poseidong.pil`,
19 committed + 18 constant columns).  Input generation only -- nothing here is measured or shipped.

  pil(nbits)            compiled PIL: the committed fixture tests/golden/poseidong.pil.json (made by tools/pilc.py from the
                        reference's source at 2^10 rows, tools/gen_poseidong_fixture.py) resized to 2^nbits rows
  consts / trace        tools/tracegen.c (semantics of starkjs/poseidon/sm_poseidong.js)
  stark_struct(nbits)   blow-up 2^ext_bits (default 2), 8 queries, FRI steps of <= 5 bits down to 2^5 (25,20,15,10,5 at
                        nBits 24); at nBits 20 (BASELINE config 3) the steps SURVEY 8 states, 21,15,11,7,4 = the mirror of
                        starky/data/r2.starkStruct.bn128.json; at nBits 10 the reference's own starkStruct, main_poseidon.js:29-39
"""
import copy
import ctypes as C
import json
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
N_CM, N_CONST = 19, 18
# the reference's two test inputs (main_poseidon.js:50-58) and their digests = the Poseidon known answers
FIRST_ZERO = [0] * 12
FIRST_COUNT = list(range(12))


def _lib():
    so, src = ROOT / "tools" / "libtracegen.so", ROOT / "tools" / "tracegen.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        import subprocess
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", str(src), "-o", str(so)])
    lib = C.CDLL(str(so))
    lib.poseidong_trace.restype = C.c_int
    return lib


def pil(nbits):
    d = json.load(open(ROOT / "tests" / "golden" / "poseidong.pil.json"))
    return resize_pil(d, nbits)


def resize_pil(d, nbits):
    """what recompiling the source with `N = 2**nbits` changes: polDeg of every reference and the row of the
    publics declared at N-1 (poseidong.pil:26-29)"""
    d = copy.deepcopy(d)
    old = next(iter(d["references"].values()))["polDeg"]
    for r in d["references"].values():
        assert r["polDeg"] == old
        r["polDeg"] = 1 << nbits
    for p in d["publics"]:
        if p["idx"] == old - 1:
            p["idx"] = (1 << nbits) - 1
        else:
            assert p["idx"] == 0
    return d


def stark_struct(nbits, n_queries=8, hash_type="GL", ext_bits=1):
    ext = nbits + ext_bits
    if nbits == 10 and ext_bits == 1:
        steps = [11, 7, 3]
    elif nbits == 20 and ext_bits == 1:
        steps = [21, 15, 11, 7, 4]
    else:
        steps, b = [ext], ext
        while b > 5:
            b = max(b - 5, 5) if b - 5 >= 4 else 4
            steps.append(b)
    return {"nBits": nbits, "nBitsExt": ext, "nQueries": n_queries, "verificationHashType": hash_type,
            "steps": [{"nBits": s} for s in steps]}


def consts(nbits):
    out = np.zeros((1 << nbits) * N_CONST, np.uint64)
    _lib().poseidong_consts(C.c_uint(nbits), out.ctypes.data_as(C.c_void_p))
    return out


def trace(nbits, n_inputs=None, first=FIRST_ZERO, seed=0):
    """n_inputs=None fills every 31-row slot with its own hash input (block 0 hashes `first`); n_inputs=1 is the
    reference's test trace: one input, the rest padded with the all-zero permutation"""
    N = 1 << nbits
    if n_inputs is None:
        n_inputs = N // 31
    out = np.zeros(N * N_CM, np.uint64)
    f = np.array(first, np.uint64)
    rc = _lib().poseidong_trace(C.c_uint(nbits), C.c_uint64(n_inputs), f.ctypes.data_as(C.c_void_p), C.c_uint64(seed),
                                out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError("Not enough Poseidon slots")
    return out


def native_program(pil_dict, ss):
    """{"starkinfo", "program"} from the product's own code generator (zk_starkinfo_generate, csrc/starkinfo_gen.hip)"""
    import importlib, sys
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import eigen_zkvm_amd
    eigen_zkvm_amd
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    return json.loads(stark.generate_program(json.dumps(pil_dict), json.dumps(ss)))


def program(nbits, ss=None):
    """{"starkinfo", "program"} of the PoseidonG PIL at 2^nbits rows"""
    return native_program(pil(nbits), ss or stark_struct(nbits))
