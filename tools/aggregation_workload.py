"""Inputs of BASELINE config 5: the three STARKs of one `test/recursive_proof_to_snark.sh` task -- the starkjs Fibonacci
circuit at 2^10 rows (:37-40, starkStruct.json.gl), a compressor-shaped circuit at 2^15 (:68-71, c12.starkStruct.json) and
at 2^18 rows (:98-102, r1.starkStruct.json) -- for any of the 8 tasks of test/stark_aggregation.sh:70-73.  The circuits are
fixed (one setup per size), the witnesses depend on the task.  Input generation only -- nothing here is measured or shipped.

The real compressor circuits are circom-compiled verifiers of the previous proof (no circom here); tools/pil/c12_shape.pil keeps
their shape: 12 committed columns, PLONK gates, a 12-column connection argument, publics at row 0."""
import ctypes as C
import json
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
FIB_INPUTS = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16)]   # fibonacci.js:63-72 has 4 pairs; 8 tasks need 8
STRUCTS = {                                                              # starky/data/{starkStruct.json.gl, c12.starkStruct.json, r1.starkStruct.json}
    "fib": {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]},
    "c12": {"nBits": 15, "nBitsExt": 16, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 16}, {"nBits": 11}, {"nBits": 7}, {"nBits": 4}]},
    "r1": {"nBits": 18, "nBitsExt": 19, "nQueries": 6, "verificationHashType": "GL", "steps": [{"nBits": 19}, {"nBits": 13}, {"nBits": 8}, {"nBits": 4}]},
}
STRUCTS["r2"] = STRUCTS["r1"]            # the joins prove recursive2 under r1.starkStruct.json too (stark_aggregation.sh:118-126)
P = 0xFFFFFFFF00000001


def _lib():
    import poseidong
    return poseidong._lib()


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def gl_root(nbits):
    """MG.0[nbits] (constant.rs:54-68)"""
    w = pow(7, 0xFFFFFFFF, P)
    for _ in range(32 - nbits):
        w = w * w % P
    return w


def c12_pil(nbits):
    """tools/pil/c12_shape.pil compiled for 2^nbits rows (tools/pilc.py)"""
    import re
    import pilc
    src = (ROOT / "tools" / "pil" / "c12_shape.pil").read_text()
    src = re.sub(r"let N: int = 2\*\*\d+;", "let N: int = 2**%d;" % nbits, src)
    return pilc.compile_pil(str(ROOT / "tools" / "pil" / "c12_shape.pil"), src)


def fib_pil():
    return json.load(open(GOLD / "starky_data" / "fib.pil.json"))


def program(kind):
    """{"starkinfo", "program"} for "fib" | "c12" | "r1" | "r2" from the product's code generator"""
    import poseidong
    return poseidong.native_program(fib_pil() if kind == "fib" else c12_pil(STRUCTS[kind]["nBits"]), STRUCTS[kind])


class Circuit:
    """one compressor-shaped circuit: constants + wiring, any number of witnesses"""

    def __init__(self, nbits, seed=12):
        self.nbits = nbits
        N = 1 << nbits
        self.consts = np.zeros(N * 26, np.uint64)
        self.wires = np.zeros(N * 8, np.uint32)
        _lib().c12s_circuit(C.c_uint(nbits), C.c_uint64(seed), C.c_uint64(gl_root(nbits)), _vp(self.consts), _vp(self.wires))

    def witness(self, task=None, primary=None):
        """the witness of task `task`, or of the given 16 primary inputs (a join: the two child roots)"""
        if primary is None:
            primary = np.random.default_rng(1000 + task).integers(0, P, size=16, dtype=np.uint64)
        primary = np.ascontiguousarray(np.asarray(primary, dtype=np.uint64) % np.uint64(P))
        assert primary.size == 16
        cm = np.zeros((1 << self.nbits) * 12, np.uint64)
        _lib().c12s_witness(C.c_uint(self.nbits), _vp(self.consts), _vp(self.wires), _vp(primary), _vp(cm))
        return cm


class JoinCircuit:
    """The join circuit of the aggregation (recursive2, test/stark_aggregation.sh:80-156) as a stand-in whose witness compressor12
    exec can compute: same PIL shape as Circuit, linear gates in layers (tools/tracegen.c c12l_*).  `.exec_text()` is its
    .exec file, `.witness_vector(primary)` what the circom calculator would hand to exec: [1, 16 primary inputs]."""
    N_WITNESS = 17

    def __init__(self, nbits, layer_bits=12, seed=13):
        self.nbits, self.layer_bits = nbits, min(layer_bits, nbits)
        N = 1 << nbits
        self.consts = np.zeros(N * 26, np.uint64)
        self.wires = np.zeros(N * 8, np.uint32)
        _lib().c12l_circuit(C.c_uint(nbits), C.c_uint(self.layer_bits), C.c_uint64(seed), C.c_uint64(gl_root(nbits)), _vp(self.consts), _vp(self.wires))

    def exec_text(self):
        f = _lib().c12l_exec_text
        f.restype = C.c_uint64
        n = f(C.c_uint(self.nbits), _vp(self.consts), _vp(self.wires), None, C.c_uint64(0))
        buf = C.create_string_buffer(n + 1)
        assert f(C.c_uint(self.nbits), _vp(self.consts), _vp(self.wires), buf, C.c_uint64(n)) == n
        return buf.raw[:n]

    @staticmethod
    def witness_vector(primary):
        v = np.zeros(17, np.uint64); v[0] = 1; v[1:] = np.asarray(primary, dtype=np.uint64) % np.uint64(P)
        return v

    def witness(self, primary):
        """the same trace by the host walk over the gates (tools/tracegen.c c12s_witness): what the exec must reproduce"""
        primary = np.ascontiguousarray(np.asarray(primary, dtype=np.uint64) % np.uint64(P))
        cm = np.zeros((1 << self.nbits) * 12, np.uint64)
        _lib().c12s_witness(C.c_uint(self.nbits), _vp(self.consts), _vp(self.wires), _vp(primary), _vp(cm))
        return cm


def fib_consts():
    out = np.zeros(2 << 10, np.uint64); _lib().fib_consts(C.c_uint(10), _vp(out)); return out


def fib_trace(task):
    a, b = FIB_INPUTS[task % len(FIB_INPUTS)]
    out = np.zeros(2 << 10, np.uint64); _lib().fib_trace(C.c_uint(10), C.c_uint64(a), C.c_uint64(b), _vp(out)); return out


def circuits():
    """{name: (constants, program_json, stark_struct_json)} of the four circuits of config 5 + the join spec ProverPool takes + the
    Circuit objects that make per-task witnesses"""
    circ = {"c12": Circuit(STRUCTS["c12"]["nBits"]), "r1": Circuit(STRUCTS["r1"]["nBits"])}
    jc = JoinCircuit(STRUCTS["r2"]["nBits"])       # the joins' circuit (recursive2): its witness is what compressor12 exec computes from 17 words
    consts = {"fib": fib_consts(), "c12": circ["c12"].consts, "r1": circ["r1"].consts, "r2": jc.consts}
    specs = {k: (consts[k], json.dumps(program(k)), json.dumps(STRUCTS[k])) for k in ("fib", "c12", "r1", "r2")}
    return specs, ("r2", jc.exec_text(), JoinCircuit.N_WITNESS, JoinCircuit.witness_vector), circ


def pool(zk, workers=4, **kw):
    """the product's ProverPool (eigen-zkvm_amd/aggregation.py) over these circuits; .task_inputs(task) makes a task's three HBM-resident traces"""
    import importlib
    agg = importlib.import_module("eigen_zkvm_amd.aggregation")
    specs, join, circ = circuits()
    p = agg.ProverPool(zk, specs, workers=workers, join=join, **kw)
    D = zk.DevArray.from_host
    p.task_inputs = lambda task: [("fib", D(fib_trace(task))), ("c12", D(circ["c12"].witness(task))), ("r1", D(circ["r1"].witness(task)))]
    p.description = ("Fibonacci 2^10 (2 columns) + compressor-shaped circuit 2^15 and 2^18 (12 columns, PLONK gates, 12-column connection; "
                     "tools/pil/c12_shape.pil), GL hash; %d tasks in flight per GPU (host threads, one stream each)" % p.workers)
    return p
