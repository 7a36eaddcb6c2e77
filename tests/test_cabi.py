"""CPU-side checks of the boundary: libzkgpu.so loads without a GPU and exports every symbol
include/zkgpu.h declares (no compute calls here)."""
import ctypes, pathlib, re
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "zkgpu.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(zk):
    lib = ctypes.CDLL(str(zk.LIB_PATH))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/zkgpu.h but not exported"
    assert sorted(zk.EXPORTS) == names


def test_constants_without_gpu(zk):
    lib = zk.lib()
    assert lib.zk_gl_modulus() == 0xFFFFFFFF00000001
    assert lib.zk_gl_root_of_unity(32) == pow(7, 2**32 - 1, 0xFFFFFFFF00000001)
    assert lib.zk_gl_root_of_unity(1) == 0xFFFFFFFF00000000
    assert lib.zk_merkle_n_nodes(256) == 511 and lib.zk_merkle_n_nodes(33) == 34 + 18 + 10 + 6 + 4 + 2 + 1
    assert lib.zk_gl_ntt_passes(24) == 3 and lib.zk_gl_ntt_passes(8) == 1 and lib.zk_gl_ntt_passes(3) == 1


def test_no_cpu_fallback_in_product():
    """The product package must not reference the oracle."""
    for f in (ROOT / "eigen-zkvm_amd").rglob("*"):
        if f.is_file() and f.suffix in (".py", ".hip", ".h", ".cpp"):
            assert "oracle" not in f.read_text().replace("no CPU fallback", ""), f


def test_host_side_argument_errors_without_gpu(zk):
    """Errors the host code raises before any device work: malformed inputs of the prover entry point, hash types,
    tree sizes -- int status / NULL + zk_last_error(), never an exception across the C ABI."""
    import json
    import numpy as np
    lib = zk.lib()
    err = lambda: lib.zk_last_error().decode()
    c = np.zeros(4, np.uint64)
    struct = {"nBits": 2, "nBitsExt": 3, "nQueries": 1, "verificationHashType": "GL", "steps": [{"nBits": 3}]}
    assert not lib.zk_stark_setup_new(b"{not json", json.dumps(struct).encode(), zk._ptr(c), 4) and "json" in err()
    assert not lib.zk_stark_setup_new(b'{"starkinfo": {}}', json.dumps(struct).encode(), zk._ptr(c), 4) and "program" in err()
    bad = dict(struct, verificationHashType="SHA256")
    assert not lib.zk_stark_setup_new(b'{"starkinfo": {}, "program": {}}', json.dumps(bad).encode(), zk._ptr(c), 4)
    assert "verificationHashType" in err()
    assert not lib.zk_stark_setup_new(None, None, None, 0) and "null" in err()
    assert lib.zk_bn128_merkle_n_nodes(256) == 256 + 16 + 1 and lib.zk_bn128_merkle_n_nodes(33) == 48 + 16 + 1
    assert lib.zk_bls12381_merkle_n_nodes(17) == 32 + 16 + 1
    assert lib.zk_bn128_load_constants(b"/nonexistent/file") != 0 and err()


def test_program_compiles_without_gpu(zk):
    """zk_program_compile needs hipRTC, not a device: the generated source is inspectable on any host"""
    p = zk.Program([zk.instr(zk.OP_MUL, zk.opnd(zk.OPND_TMP, id=0, dim=1), zk.opnd(zk.OPND_NUMBER, value=3), zk.opnd(zk.OPND_NUMBER, value=5))])
    src = zk.lib().zk_program_source(p._h)
    assert src and b"zk_eval_kernel" in src


def test_groth16_and_compressor_readers_without_gpu(zk):
    """the file readers run on the host: the reference's .wtns fixture (test/single/witness.wtns) and the header errors of
    reader.rs:86-137 / r1cs_file.rs:185-200 / compressor12_exec.rs:110-125 surface before any device work"""
    import importlib
    dev = importlib.import_module("eigen_zkvm_amd.groth16")
    b = (ROOT / "tests" / "golden" / "groth16" / "witness.wtns").read_bytes()
    w = dev.wtns_values(b, "BN128")
    assert w[:, 0].tolist() == [1, 11210000, 1121, 10000] and not w[:, 1:].any()
    for bad, msg in ((b"wtnx" + b[4:], "Invalid file header"), (b[:8] + b"\x03\x00\x00\x00" + b[12:], "invalid num sections"), (b[:-1], "truncated")):
        with pytest.raises(zk.ZkError, match=msg):
            dev.wtns_values(bad, "BN128")
    with pytest.raises(zk.ZkError, match="invalid curve prime"):
        dev.wtns_values(b, "BLS12381")
    lib = zk.lib(); err = lambda: lib.zk_last_error().decode()
    r1 = (ROOT / "tests" / "golden" / "groth16" / "mycircuit_bls12381.r1cs").read_bytes()
    assert not lib.zk_groth16_setup_new(b"BN128", r1, len(r1), b"x", 1) and "prime is not the scalar field" in err()
    assert not lib.zk_groth16_setup_new(b"BLS12381", b"r1cx" + r1[4:], len(r1), b"x", 1) and "Invalid magic number" in err()
    assert not lib.zk_groth16_setup_new(b"BLS12381", r1, len(r1), b"x", 1) and "truncated" in err()       # the circuit parses; the key does not
    assert not lib.zk_c12_exec_new(b"[1,0,5,6,7]", 11, 10) and "length does not match" in err()
    assert not lib.zk_c12_exec_new(b"[1,0,12,0,1,1]", 14, 10) and "does not exist yet" in err()


def test_stark_verify_rejects_malformed_input_before_touching_a_device(zk):
    """zk_stark_verify_with: -1 (an error, with a message) for unparseable or incomplete input -- decided by the host parser, so it can
    be checked without a GPU; 0 / 1 need the device's hashes (tests/test_gpu_verify.py)"""
    import ctypes as C, json
    lib = zk.lib()
    root = (C.c_uint64 * 4)(1, 2, 3, 4)
    ss = json.dumps({"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]})
    prog = json.dumps({"starkinfo": {"q_deg": 1, "qs": [0], "ev_idx": {"cm": [], "const_": []}}, "program": {"verifier_code": {"first": []}, "verifier_query_code": {"first": []}}})
    for args in ((b"{", ss.encode(), b"{}"), (prog.encode(), b"not json", b"{}"), (prog.encode(), ss.encode(), b"{"),
                 (prog.encode(), ss.encode(), b"{}"),                                     # no root1
                 (prog.encode(), json.dumps(dict(json.loads(ss), verificationHashType="SHA256")).encode(), b"{}"),
                 (json.dumps({"program": {}}).encode(), ss.encode(), b"{}")):
        assert lib.zk_stark_verify_with(args[0], args[1], root, args[2]) == -1
        assert lib.zk_last_error()
    assert lib.zk_stark_verify_with(None, ss.encode(), root, b"{}") == -1 and b"null" in lib.zk_last_error()
    assert lib.zk_stark_verify(None, b"{}") == -1


def test_poseidon_matrix_pipe_tables_selfcheck():
    """Round 6: the one-lane Poseidon kernels compute their dense 64-bit products on the matrix pipe from digit tables built on the host
    (csrc/gl_mfma.hip.h).  zk_gl_poseidon_selfcheck needs no GPU: every table row must spell c 2^(8 b) mod p, the fragments must sit where the
    kernel reads them, and the pipe's arithmetic -- i32 columns of (byte - 128) x digit, two biased 64-bit sums, the fold -- replayed on the host
    must equal 128-bit arithmetic for P and the four block products on random and extreme vectors (0, 2^64 - 1, p - 1, 0x80..80, 0x7F..7F)."""
    import zkgpu_loader
    zk = zkgpu_loader.load()
    assert zk.lib().zk_gl_poseidon_selfcheck() == 0, zk.lib().zk_last_error().decode()


@pytest.mark.parametrize("field", ["bn128", "bls12381"])
def test_scalar_field_poseidon_matrix_pipe_tables_selfcheck(field):
    """Round 6: the dense layers (M, P) of the one-lane scalar-field Poseidon kernels run on the matrix pipe from digit tables built on the
    host when the constants are loaded (csrc/fr_mfma.hip.h).  zk_<field>_poseidon_selfcheck needs no GPU: for every t = 3..17 and both
    matrices it builds the tables from the constants file and replays the device's arithmetic in host integers -- i32 columns of
    (byte - 128) x digit, eight biased 64-bit words, the Montgomery step -- on random and extreme vectors (2^256 - 1, 0, 0x80..80, 0x7F..7F);
    every output must be congruent to sum_j c x_j + addend, below 2r, with every intermediate inside the range the device code assumes."""
    import zkgpu_loader
    zk = zkgpu_loader.load()
    zk.bn128_tables_selfcheck(field=field)

