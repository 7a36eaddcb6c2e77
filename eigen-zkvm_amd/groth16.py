"""Host-side mirror of the reference's Groth16 entry points over libzkgpu's C ABI (include/zkgpu.h, "Groth16 around
the multi-scalar sums"): groth16/src/api.rs:144-205 groth16_prove = read the key, read the circuit, read the
witness, create_random_proof, serialize_proof.  Everything after the witness is on the device; there is no CPU
fallback."""
import ctypes as C
import json
import secrets

import numpy as np

from . import DevArray, ZkError, _check, _np, _ptr, lib

_FR = {"BN128": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
       "BLS12381": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001}
_NAME = {"BN128": "bn254", "BLS12381": "bls12_381"}
_FQ_WORDS = {"BN128": 4, "BLS12381": 6}


def fr_ntt(data, curve="BN128", inverse=False, coset=False):
    """EvaluationDomain::{fft, ifft, coset_fft, icoset_fft} on an n x 4 u64 array of Montgomery Fr limbs (n = 2^k);
    a DevArray is transformed in place, a host array is copied and returned"""
    fn = "zk_fr_%s_ntt" % _NAME[curve]
    if isinstance(data, DevArray):
        n = data.n // 4
        _check(getattr(lib(), fn + "_dev")(data.ptr, n.bit_length() - 1, int(inverse), int(coset), 0)); return data
    a = _np(data).reshape(-1).copy()
    n = a.size // 4
    if n == 0 or n & (n - 1):
        raise ZkError("fr ntt: length must be a power of two")
    _check(getattr(lib(), fn)(_ptr(a), n.bit_length() - 1, int(inverse), int(coset))); return a.reshape(-1, 4)


def fr_quotient(d_a, d_b, d_c, curve="BN128"):
    """create_proof's h block on three DevArrays of 2^k Montgomery Fr elements; d_a is overwritten with the coefficients"""
    n = d_a.n // 4
    _check(getattr(lib(), "zk_fr_%s_quotient_dev" % _NAME[curve])(d_a.ptr, d_b.ptr, d_c.ptr, n.bit_length() - 1, 0)); return d_a


def wtns_values(wtns_bytes, curve="BN128"):
    """load_witness_from_bin_reader (algebraic/src/reader.rs:86-137): the n x 4 u64 canonical values of a .wtns file"""
    off, n = C.c_uint64(0), C.c_uint64(0)
    buf = np.frombuffer(wtns_bytes, dtype=np.uint8)
    _check(lib().zk_groth16_wtns_payload(buf.ctypes.data, buf.size, curve.encode(), C.byref(off), C.byref(n)))
    return np.frombuffer(wtns_bytes, dtype="<u8", count=4 * n.value, offset=off.value).reshape(-1, 4).copy()


class Groth16Setup:
    """The state groth16_prove rebuilds per call (api.rs:161-171), kept resident: proving key as device bases, the
    circuit's three CSR matrices, the density index lists and the transform tables of its domain."""

    def __init__(self, curve, r1cs_bytes, params_bytes):
        if curve not in _FR:
            raise ZkError('groth16: unknown curve "%s" (BN128 | BLS12381)' % curve)
        self.curve = curve
        r = np.frombuffer(r1cs_bytes, dtype=np.uint8); p = np.frombuffer(params_bytes, dtype=np.uint8)
        self._h = lib().zk_groth16_setup_new(curve.encode(), r.ctypes.data, r.size, p.ctypes.data, p.size)
        if not self._h:
            raise ZkError(lib().zk_last_error().decode())
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _check(lib().zk_groth16_setup_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.n_wires, self.n_inputs, self.domain_log = a.value, b.value, c.value

    def prove(self, witness, r=None, s=None, d_h=None):
        """witness: n_wires x 4 u64 canonical (host array or DevArray); r, s: blinding scalars (drawn here when None,
        as create_random_proof does).  -> (proof.json dict, points: A || B || C u64 Montgomery words)"""
        mod = _FR[self.curve]
        r = secrets.randbelow(mod) if r is None else r
        s = secrets.randbelow(mod) if s is None else s
        if not (0 <= r < mod and 0 <= s < mod):
            raise ZkError("groth16: r and s must be canonical field elements")
        rw = np.array([(r >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        sw = np.array([(s >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        pts = np.zeros(8 * _FQ_WORDS[self.curve], np.uint64)
        if isinstance(witness, DevArray):
            p = lib().zk_groth16_prove_dev(self._h, witness.ptr, witness.n // 4, _ptr(rw), _ptr(sw), _ptr(pts), d_h.ptr if d_h is not None else None)
        else:
            w = _np(witness).reshape(-1)
            p = lib().zk_groth16_prove(self._h, _ptr(w), w.size // 4, _ptr(rw), _ptr(sw), _ptr(pts))
        if not p:
            raise ZkError(lib().zk_last_error().decode())
        try:
            return json.loads(C.string_at(p).decode()), pts
        finally:
            lib().zk_string_free(p)

    def free(self):
        if self._h:
            lib().zk_groth16_setup_free(self._h); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def fq_convert(d_elems, curve="BN128", to_mont=True, stream=0):
    """Fq::from_repr / into_repr on a DevArray of base-field elements, in place"""
    nl = _FQ_WORDS[curve]
    _check(getattr(lib(), "zk_fq_%s_convert_dev" % _NAME[curve])(d_elems.ptr, d_elems.n // nl, int(to_mont), stream)); return d_elems
