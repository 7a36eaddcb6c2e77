"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (the checker, never the
product).  Builds the oracle on first use if the .so is missing (gcc only, seconds)."""
import ctypes as C
import pathlib, subprocess
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
P = 0xFFFFFFFF00000001
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


class Oracle:
    def threads(self):
        return int(self.lib.orc_threads())

    def __init__(self, lib):
        self.lib = lib
        L = lib
        for name in ("orc_gl_mul", "orc_gl_mul_slow", "orc_gl_add", "orc_gl_sub", "orc_gl_pow"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_uint64, C.c_uint64]
        L.orc_gl_inv.restype = C.c_uint64; L.orc_gl_inv.argtypes = [C.c_uint64]
        L.orc_gl_root.restype = C.c_uint64; L.orc_gl_root.argtypes = [C.c_uint]
        L.orc_bitrev.restype = C.c_uint32; L.orc_bitrev.argtypes = [C.c_uint32, C.c_uint]
        L.orc_threads.restype = C.c_int; L.orc_threads.argtypes = []
        L.orc_f3_mul.argtypes = [_u64p, _u64p, _u64p]
        L.orc_f3_inv.argtypes = [_u64p, _u64p]
        L.orc_f3_pow.argtypes = [_u64p, C.c_uint64, _u64p]
        L.orc_ntt.argtypes = [_u64p, _u64p, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_ntt_blocked.argtypes = [_u64p, _u64p, C.c_uint32, C.c_int]
        L.orc_lde.argtypes = [_u64p, C.c_uint32, C.c_uint32, _u64p, C.c_uint32]
        L.orc_poseidon.argtypes = [_u64p, _u64p, _u64p, C.c_int]
        L.orc_linearhash.argtypes = [_u64p, C.c_size_t, _u64p]
        L.orc_merkle_n_nodes.restype = C.c_uint64; L.orc_merkle_n_nodes.argtypes = [C.c_uint64]
        L.orc_merkelize.argtypes = [_u64p, C.c_uint32, C.c_uint64, _u64p]
        L.orc_merkle_proof.restype = C.c_int
        L.orc_merkle_proof.argtypes = [_u64p, C.c_uint64, C.c_uint64, _u64p]
        L.orc_merkle_root_from_proof.argtypes = [_u64p, C.c_uint32, _u64p, C.c_int, C.c_uint64, _u64p]
        L.orc_tr_sizeof.restype = C.c_size_t
        L.orc_tr_init.argtypes = [C.c_void_p]
        L.orc_tr_put.argtypes = [C.c_void_p, _u64p, C.c_size_t]
        L.orc_tr_get1.restype = C.c_uint64; L.orc_tr_get1.argtypes = [C.c_void_p]
        L.orc_tr_get_field.argtypes = [C.c_void_p, _u64p]
        L.orc_tr_get_permutations.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _u64p]
        L.orc_f3_ntt.argtypes = [_u64p, C.c_uint, C.c_int]
        L.orc_fri_fold.argtypes = [_u64p, C.c_uint, C.c_uint, _u64p, C.c_uint64, _u64p]
        L.orc_fri_transpose.argtypes = [_u64p, C.c_uint64, C.c_uint, _u64p]
        L.orc_f3_batch_inverse.argtypes = [_u64p, C.c_uint64, _u64p]
        L.orc_xdivxsub.argtypes = [_u64p, C.c_uint, _u64p]
        L.orc_zh_inv.argtypes = [C.c_uint, C.c_uint, _u64p]
        L.orc_lev.argtypes = [_u64p, C.c_uint, C.c_int, _u64p]
        L.orc_eval_dot.argtypes = [_u64p, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint, C.c_uint, _u64p, _u64p]
        L.orc_qsplit.argtypes = [_u64p, C.c_uint, C.c_uint, C.c_uint, _u64p]
        L.orc_calculate_z.restype = C.c_int
        L.orc_calculate_z.argtypes = [_u64p, _u64p, C.c_uint64, _u64p]

    # -- field
    def mul(self, a, b): return self.lib.orc_gl_mul(a, b)
    def add(self, a, b): return self.lib.orc_gl_add(a, b)
    def sub(self, a, b): return self.lib.orc_gl_sub(a, b)
    def inv(self, a): return self.lib.orc_gl_inv(a)
    def pow(self, a, e): return self.lib.orc_gl_pow(a, e)
    def root(self, k): return self.lib.orc_gl_root(k)
    def f3_mul(self, a, b):
        o = np.zeros(3, np.uint64); self.lib.orc_f3_mul(_a(a), _a(b), o); return o
    def f3_inv(self, a):
        o = np.zeros(3, np.uint64); self.lib.orc_f3_inv(_a(a), o); return o
    def f3_pow(self, a, e):
        o = np.zeros(3, np.uint64); self.lib.orc_f3_pow(_a(a), e, o); return o

    # -- transforms (row-major [1<<nbits][n_pols])
    def ntt(self, src, n_pols, nbits, inverse=False):
        src = _a(src); dst = np.empty_like(src)
        self.lib.orc_ntt(src, dst, n_pols, nbits, int(inverse)); return dst
    def ntt_blocked(self, src, nbits, inverse=False):
        """one column, all host threads, blocks + transposes like fft_p.rs -- bench.py's cpu_baseline"""
        src = _a(src); dst = np.empty_like(src)
        self.lib.orc_ntt_blocked(src, dst, nbits, int(inverse)); return dst
    def lde(self, src, n_pols, nbits, nbits_ext):
        src = _a(src); dst = np.empty((1 << nbits_ext) * n_pols, np.uint64)
        self.lib.orc_lde(src, n_pols, nbits, dst, nbits_ext); return dst

    # -- hashing
    def poseidon(self, inp, cap, n_out=4):
        o = np.zeros(n_out, np.uint64); self.lib.orc_poseidon(_a(inp), _a(cap), o, n_out); return o
    def linearhash(self, v):
        v = _a(v); o = np.zeros(4, np.uint64); self.lib.orc_linearhash(v, v.size, o); return o
    def n_nodes(self, height): return self.lib.orc_merkle_n_nodes(height)
    def merkelize(self, buff, width, height):
        nodes = np.zeros(self.n_nodes(height) * 4, np.uint64)
        self.lib.orc_merkelize(_a(buff), width, height, nodes); return nodes
    def merkle_proof(self, nodes, height, idx):
        path = np.zeros(64 * 4, np.uint64)
        d = self.lib.orc_merkle_proof(nodes, height, idx, path); return path[:4 * d].copy()
    def root_from_proof(self, row, path, idx):
        row = _a(row); path = _a(path); r = np.zeros(4, np.uint64)
        self.lib.orc_merkle_root_from_proof(row, row.size, path, path.size // 4, idx, r); return r

    def transcript(self):
        return Transcript(self)

    # -- BN254 G1 (oracle/ec.c); points = 8 u64 words x||y (Montgomery), scalars = 4 words LE
    def _ec_setup(self):
        L = self.lib
        if getattr(self, "_ec_ready", False):
            return
        L.orc_bn254_generator.argtypes = [_u64p]
        L.orc_bn254_on_curve.restype = C.c_int; L.orc_bn254_on_curve.argtypes = [_u64p]
        L.orc_fq_mul.argtypes = [_u64p, _u64p, _u64p]
        L.orc_fq_from_mont.argtypes = [_u64p, _u64p]; L.orc_fq_to_mont.argtypes = [_u64p, _u64p]
        L.orc_bn254_scalar_mul.restype = C.c_int; L.orc_bn254_scalar_mul.argtypes = [_u64p, _u64p, _u64p]
        L.orc_bn254_make_bases.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u64p]
        L.orc_bn254_msm.restype = C.c_int; L.orc_bn254_msm.argtypes = [_u64p, _u64p, C.c_uint64, C.c_uint, _u64p]
        self._ec_ready = True
    def bn254_generator(self):
        self._ec_setup(); o = np.zeros(8, np.uint64); self.lib.orc_bn254_generator(o); return o
    def bn254_on_curve(self, p):
        self._ec_setup(); return bool(self.lib.orc_bn254_on_curve(_a(p)))
    def fq_from_mont(self, a):
        self._ec_setup(); o = np.zeros(4, np.uint64); self.lib.orc_fq_from_mont(_a(a), o); return o
    def bn254_scalar_mul(self, p, k):
        self._ec_setup(); o = np.zeros(8, np.uint64)
        inf = self.lib.orc_bn254_scalar_mul(_a(p), _a(k), o); return o, bool(inf)
    def bn254_make_bases(self, n, a, b):
        self._ec_setup(); o = np.zeros(8 * n, np.uint64); self.lib.orc_bn254_make_bases(n, a, b, o); return o
    def bn254_msm(self, bases, scalars, c=8):
        self._ec_setup(); bases = _a(bases); o = np.zeros(8, np.uint64)
        inf = self.lib.orc_bn254_msm(bases, _a(scalars), bases.size // 8, c, o); return o, bool(inf)

    def bn128(self):
        """BN128-field Poseidon / LinearHash / Merkle / transcript (oracle/bn128_hash.c)"""
        return BN128Hash(self.lib, "bn128")

    def bls12381(self):
        """the same over the BLS12-381 scalar field (oracle/bls12381_hash.c)"""
        return BN128Hash(self.lib, "bls12381")

    def curve(self, name, g2=False):
        """G1 (or, g2=True, G2) arithmetic of `name` in ("bn254", "bls12_381") (oracle/ec.c, ec_bls12_381.c over ec_impl.h)."""
        return Curve(self.lib, name, g2)

    # -- prover glue (oracle/stark_steps.c)
    def f3_ntt(self, v, bits, inverse=False):
        v = _a(v).copy(); self.lib.orc_f3_ntt(v, bits, int(inverse)); return v
    def fri_fold(self, pol, pol_bits, step_bits, special_x, shift_inv):
        o = np.zeros(3 << step_bits, np.uint64)
        self.lib.orc_fri_fold(_a(pol), pol_bits, step_bits, _a(special_x), shift_inv, o); return o
    def fri_transpose(self, pol, n, tbits):
        o = np.zeros(3 * n, np.uint64); self.lib.orc_fri_transpose(_a(pol), n, tbits, o); return o
    def f3_batch_inverse(self, v):
        v = _a(v); o = np.zeros_like(v); self.lib.orc_f3_batch_inverse(v, v.size // 3, o); return o
    def xdivxsub(self, xi, nbits_ext):
        o = np.zeros(3 << nbits_ext, np.uint64); self.lib.orc_xdivxsub(_a(xi), nbits_ext, o); return o
    def zh_inv(self, nbits, ext):
        o = np.zeros(1 << ext, np.uint64); self.lib.orc_zh_inv(nbits, ext, o); return o
    def lev(self, xi, nbits, prime):
        o = np.zeros(3 << nbits, np.uint64); self.lib.orc_lev(_a(xi), nbits, int(prime), o); return o
    def eval_dot(self, buf, width, offset, dim, nbits, ext, L):
        o = np.zeros(3, np.uint64); self.lib.orc_eval_dot(_a(buf), width, offset, dim, nbits, ext, _a(L), o); return o
    def qsplit(self, qq1, nbits, nbits_ext, q_dim, q_deg):
        o = np.zeros((1 << nbits_ext) * q_dim * q_deg, np.uint64)
        self.lib.orc_qsplit(_a(qq1), nbits, q_dim, q_deg, o); return o
    def calculate_z(self, num, den):
        num = _a(num); z = np.zeros_like(num)
        ok = self.lib.orc_calculate_z(num, _a(den), num.size // 3, z); return z, bool(ok)


class Transcript:
    def __init__(self, o):
        self.o = o; self.buf = C.create_string_buffer(o.lib.orc_tr_sizeof())
        o.lib.orc_tr_init(self.buf)
    def put(self, v):
        v = _a(v); self.o.lib.orc_tr_put(self.buf, v, v.size)
    def get1(self): return self.o.lib.orc_tr_get1(self.buf)
    def get_field(self):
        o = np.zeros(3, np.uint64); self.o.lib.orc_tr_get_field(self.buf, o); return o
    def get_permutations(self, n, nbits):
        o = np.zeros(n, np.uint64); self.o.lib.orc_tr_get_permutations(self.buf, n, nbits, o); return o


def _a(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64).reshape(-1))


def usable_cpus():
    """CPUs this process can keep busy: the affinity mask capped by the cgroup CPU quota (cpu.max / cfs_quota_us)"""
    import math, os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / per)))
        except (OSError, ValueError):
            pass
    return n


def build():
    subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle")])


_cached = None
class BN128Hash:
    """digests and field elements travel as 4 u64 raw (Montgomery) limbs, like ElementDigest<4, Fr>"""
    MODULI = {"bn128": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
              "bls12381": 52435875175126190479447740508185965837690552500527637822603658699938581184513}

    class _Lib:
        """view of liboracle.so that maps orc_bn128_x to orc_<field>_x"""
        def __init__(self, lib, field):
            self._lib, self._field = lib, field
        def __getattr__(self, name):
            return getattr(self._lib, name.replace("orc_bn128_", "orc_%s_" % self._field))

    def __init__(self, lib, field="bn128"):
        self.field, self.R = field, self.MODULI[field]
        self.lib = L = self._Lib(lib, field)
        L.orc_bn128_load_constants.argtypes = [C.c_char_p]; L.orc_bn128_load_constants.restype = C.c_int
        path = str(pathlib.Path(__file__).resolve().parent.parent / "eigen-zkvm_amd" / "data" / ("poseidon_%s_constants.bin" % field))
        assert L.orc_bn128_load_constants(path.encode()) == 0, "cannot load " + path
        L.orc_bn128_hash.argtypes = [_u64p, C.c_uint32, _u64p, _u64p]; L.orc_bn128_hash.restype = C.c_int
        L.orc_bn128_fr_to_mont.argtypes = [_u64p, _u64p]; L.orc_bn128_fr_from_mont.argtypes = [_u64p, _u64p]
        L.orc_bn128_poseidon.argtypes = [_u64p, C.c_uint32, _u64p, C.c_uint32, _u64p]; L.orc_bn128_poseidon.restype = C.c_int
        L.orc_bn128_hash_element_array.argtypes = [_u64p, C.c_uint64, _u64p]; L.orc_bn128_hash_element_array.restype = C.c_int
        L.orc_bn128_hash_element_matrix.argtypes = [_u64p, C.c_uint64, _u64p]; L.orc_bn128_hash_element_matrix.restype = C.c_int
        L.orc_bn128_merkle_n_nodes.argtypes = [C.c_uint64]; L.orc_bn128_merkle_n_nodes.restype = C.c_uint64
        L.orc_bn128_merkle_depth.argtypes = [C.c_uint64]; L.orc_bn128_merkle_depth.restype = C.c_uint32
        L.orc_bn128_merkelize.argtypes = [_u64p, C.c_uint32, C.c_uint64, _u64p]; L.orc_bn128_merkelize.restype = C.c_int
        L.orc_bn128_merkle_proof.argtypes = [_u64p, C.c_uint64, C.c_uint64, _u64p]; L.orc_bn128_merkle_proof.restype = None
        L.orc_bn128_merkle_root_from_proof.argtypes = [_u64p, C.c_uint32, _u64p, _u64p]; L.orc_bn128_merkle_root_from_proof.restype = C.c_int
        L.orc_bn128_tr_new.restype = C.c_void_p; L.orc_bn128_tr_free.argtypes = [C.c_void_p]
        L.orc_bn128_tr_put1.argtypes = [C.c_void_p, C.c_uint64]; L.orc_bn128_tr_put4.argtypes = [C.c_void_p, _u64p]
        L.orc_bn128_tr_get_fields1.argtypes = [C.c_void_p, _u64p]
        L.orc_bn128_tr_get_permutations.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _u64p]

    @staticmethod
    def words(x):
        return np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)
    @staticmethod
    def to_int(w):
        return sum(int(v) << (64 * i) for i, v in enumerate(w))
    def to_mont(self, x):
        """canonical integer -> raw limbs"""
        o = np.zeros(4, np.uint64); self.lib.orc_bn128_fr_to_mont(self.words(x % self.R), o); return o
    def from_mont(self, raw):
        """raw limbs -> canonical integer"""
        o = np.zeros(4, np.uint64); self.lib.orc_bn128_fr_from_mont(_a(raw), o); return self.to_int(o)
    def poseidon(self, inp_raw, init_raw=None, n_out=1):
        inp = _a(np.asarray(inp_raw, np.uint64).reshape(-1)); n = inp.size // 4
        init = _a(init_raw) if init_raw is not None else np.zeros(4, np.uint64)
        o = np.zeros(4 * n_out, np.uint64)
        assert self.lib.orc_bn128_poseidon(inp, n, init, n_out, o) == 0
        return o.reshape(n_out, 4)
    def hash_ints(self, vals, init=0, n_out=1):
        """Poseidon::hash_ex over canonical integers -> canonical integers"""
        raw = np.concatenate([self.to_mont(v) for v in vals])
        return [self.from_mont(r) for r in self.poseidon(raw, self.to_mont(init), n_out)]
    def hash1_ints(self, vals, init=0):
        """Poseidon::hash (one element: state[0] for BN128, state[1] for BLS12-381), the form of the reference's tests"""
        raw = np.concatenate([self.to_mont(v) for v in vals]); o = np.zeros(4, np.uint64)
        assert self.lib.orc_bn128_hash(raw, len(vals), self.to_mont(init), o) == 0
        return self.from_mont(o)
    def hash_element_array(self, vals):
        v = _a(vals); o = np.zeros(4, np.uint64); assert self.lib.orc_bn128_hash_element_array(v, v.size, o) == 0; return o
    def hash_element_matrix(self, vals):
        v = _a(vals); o = np.zeros(4, np.uint64); assert self.lib.orc_bn128_hash_element_matrix(v, v.size, o) == 0; return o
    def n_nodes(self, height):
        return int(self.lib.orc_bn128_merkle_n_nodes(height))
    def depth(self, height):
        return int(self.lib.orc_bn128_merkle_depth(height))
    def merkelize(self, rows, width, height):
        r = _a(rows); nodes = np.zeros(4 * self.n_nodes(height), np.uint64)
        assert self.lib.orc_bn128_merkelize(r, width, height, nodes) == 0
        return nodes.reshape(-1, 4)
    def merkle_proof(self, nodes, height, idx):
        path = np.zeros(self.depth(height) * 64, np.uint64)
        self.lib.orc_bn128_merkle_proof(_a(nodes.reshape(-1)), height, idx, path); return path.reshape(-1, 16, 4)
    def root_from_proof(self, path, leaf):
        o = np.zeros(4, np.uint64)
        assert self.lib.orc_bn128_merkle_root_from_proof(_a(path.reshape(-1)), path.shape[0], _a(leaf), o) == 0; return o
    def transcript(self):
        return BN128Transcript(self)


class BN128Transcript:
    def __init__(self, h):
        self.h, self.p = h, h.lib.orc_bn128_tr_new()
    def put1(self, v):
        assert self.h.lib.orc_bn128_tr_put1(self.p, int(v)) == 0
    def put4(self, d):
        assert self.h.lib.orc_bn128_tr_put4(self.p, _a(d)) == 0
    def get_fields1(self):
        o = np.zeros(1, np.uint64); assert self.h.lib.orc_bn128_tr_get_fields1(self.p, o) == 0; return int(o[0])
    def get_field(self):
        return [self.get_fields1() for _ in range(3)]
    def get_permutations(self, n, nbits):
        o = np.zeros(n, np.uint64); assert self.h.lib.orc_bn128_tr_get_permutations(self.p, n, nbits, o) == 0; return o
    def __del__(self):
        try:
            self.h.lib.orc_bn128_tr_free(self.p)
        except Exception:
            pass


class Curve:
    """points = 2*nl u64 words x||y (Montgomery), scalars = 4 words canonical little-endian"""
    PARAMS = {"bn254": (4, 21888242871839275222246405745257275088548364400416034343698204186575808495617),
              "bls12_381": (6, 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001)}

    def __init__(self, lib, name, g2=False):
        """g2: the same interface over the twist (points = 4*nl words x.c0 || x.c1 || y.c0 || y.c1)"""
        self.nl, self.r = self.PARAMS[name]
        self.name, self.g2 = name, g2
        self.pw = (4 if g2 else 2) * self.nl                       # words per affine point
        f = lambda n: getattr(lib, "orc_%s_%s%s" % (name, "g2_" if g2 and not n.startswith("fq_") else "", n))
        self._gen, self._on, self._mul, self._bases, self._msm, self._from_mont = (
            f("generator"), f("on_curve"), f("scalar_mul"), f("make_bases"), f("msm"), f("fq_from_mont"))
        self._gen.argtypes = [_u64p]; self._gen.restype = None
        self._on.argtypes = [_u64p]; self._on.restype = C.c_int
        self._mul.argtypes = [_u64p, _u64p, _u64p]; self._mul.restype = C.c_int
        self._bases.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u64p]; self._bases.restype = None
        self._msm.argtypes = [_u64p, _u64p, C.c_uint64, C.c_uint, _u64p]; self._msm.restype = C.c_int
        self._from_mont.argtypes = [_u64p, _u64p]; self._from_mont.restype = None
        if not g2:
            self._msm_par = getattr(lib, "orc_%s_msm_par" % name)
            self._msm_par.argtypes = [_u64p, _u64p, C.c_uint64, C.c_uint, C.c_uint, _u64p]; self._msm_par.restype = C.c_int

    def generator(self):
        o = np.zeros(self.pw, np.uint64); self._gen(o); return o
    def on_curve(self, p):
        return bool(self._on(_a(p)))
    def fq_from_mont(self, a):
        o = np.zeros(self.nl, np.uint64); self._from_mont(_a(a), o); return o
    def scalar_mul(self, p, k):
        o = np.zeros(self.pw, np.uint64); inf = self._mul(_a(p), _a(k), o); return o, bool(inf)
    def make_bases(self, n, a, b):
        o = np.zeros(self.pw * n, np.uint64); self._bases(n, a, b, o); return o
    def msm(self, bases, scalars, c=8):
        bases = _a(bases); o = np.zeros(self.pw, np.uint64)
        inf = self._msm(bases, _a(scalars), bases.size // self.pw, c, o); return o, bool(inf)
    def msm_par(self, bases, scalars, c=8, chunks=1):
        """the same sum with (window, chunk) tasks over every host thread -- bench.py's cpu_baseline"""
        bases = _a(bases); o = np.zeros(self.pw, np.uint64)
        inf = self._msm_par(bases, _a(scalars), bases.size // self.pw, c, chunks, o); return o, bool(inf)
    def affine_ints(self, p):
        """canonical integers of the coordinates: (x, y) for G1, (x.c0, x.c1, y.c0, y.c1) for G2"""
        to_int = lambda w: sum(int(v) << (64 * i) for i, v in enumerate(w))
        return tuple(to_int(self.fq_from_mont(p[i * self.nl:(i + 1) * self.nl])) for i in range(self.pw // self.nl))


def load():
    global _cached
    if _cached is None:
        so = ROOT / "oracle" / "liboracle.so"
        if not so.exists():
            build()
        lib = C.CDLL(str(so))
        if "OMP_NUM_THREADS" not in __import__("os").environ:        # an explicit setting wins
            lib.orc_set_threads(C.c_int(usable_cpus()))
        _cached = Oracle(lib)
    return _cached


def splitmix64_stream(seed, n):
    """x_i = splitmix64(seed, i) mod p -- the synthetic input of BASELINE config 2."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return np.where(z >= np.uint64(P), z - np.uint64(P), z)
