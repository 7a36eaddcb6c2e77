"""Whole proofs at size, through the C++ driver (zk_stark_setup_new / zk_stark_gen_dev), against the oracle:

  * BASELINE config 3's PIL (starkjs/poseidon/poseidong.pil: 19 committed + 18 constant columns, 36 intermediate
    columns in cm3, q_deg 2) at 2^10 (the reference's own size, main_poseidon.js:29-39) and 2^16 rows: zkin **equal**
    to the oracle prover's, byte for byte; at 2^20 rows (config 3 itself): accepted by the restated verifier
    (stark_verify.rs:20-136), a tampered copy rejected -- what the reference's own end-to-end tests assert
    (stark_gen.rs:981-1195);
  * the reference's connection PIL scaled to 2^16 rows, so that calculate_Z (stark_gen.rs:653-666) runs as a multi-block
    scan inside a proof: zkin equal to the oracle's;
  * unit checks at sizes the 2^10 fixtures never reach: the grand product against orc_calculate_z up to 2^20, the
    2^24 -> 2^25 extension (4-pass plan, tw_mid / dshift paths) by linearity, a decimated oracle comparison and the
    inverse round trip.
"""
import copy
import json
import pathlib
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tools"))
P = 0xFFFFFFFF00000001


def _stark(zk):
    import importlib
    zk.init(0)
    return importlib.import_module("eigen_zkvm_amd.stark")


def _native(stark, const, info_prog_json, ss):
    return stark.NativeStarkSetup(const, info_prog_json, json.dumps(ss))


@pytest.mark.parametrize("nbits,n_inputs", [(10, 1), (10, None), (13, None), (16, None)])
def test_poseidong_zkin_equals_oracle(zk, orc, nbits, n_inputs):
    import stark_prover as SP, starkinfo as SI, poseidong as PG
    stark = _stark(zk)
    ss, const = PG.stark_struct(nbits), PG.consts(nbits)
    cm = PG.trace(nbits, n_inputs, PG.FIRST_COUNT, seed=nbits)
    su = SP.setup(PG.pil(nbits), const, ss, orc)
    info = su["starkinfo"]
    assert (info["n_cm1"], info["n_constants"], info["n_cm3"], info["q_deg"], info["q_dim"]) == (19, 18, 36, 2, 3)
    exp = SP.to_zkin(SP.stark_gen(cm, su, ss, orc))
    # the product's own code generator (what the bench uses) makes what the oracle's makes
    program = PG.program(nbits)
    assert program == json.loads(json.dumps(SI.to_json(su["starkinfo"], su["program"])))
    ns = _native(stark, const, json.dumps(program), ss)
    got = ns.gen(zk.DevArray.from_host(cm))
    assert list(got) == list(exp)
    for k in exp:
        assert got[k] == exp[k], k
    assert got == ns.gen(cm)                                                  # host-trace entry point, same bytes
    ns.free()


def test_poseidong_2p20_verifies_and_tamper_rejected(zk, orc):
    """BASELINE config 3: 2^20-row PoseidonG proof, every slot hashing its own input"""
    import stark_prover as SP, starkinfo as SI, poseidong as PG
    stark = _stark(zk)
    nbits = 20
    ss, const = PG.stark_struct(nbits), PG.consts(nbits)
    assert [s["nBits"] for s in ss["steps"]] == [21, 15, 11, 7, 4]                            # the mirror of r2.starkStruct.bn128.json
    cm = PG.trace(nbits, None, PG.FIRST_ZERO, seed=20)
    info, prog, _ = SI.generate(PG.pil(nbits), ss)
    ns = _native(stark, const, json.dumps(PG.program(nbits)), ss)
    z = ns.gen(zk.DevArray.from_host(cm))
    root_c = [str(v) for v in ns.const_root()]
    ns.free()
    assert z["rootC"] == root_c
    assert z["publics"][:8] == ["0"] * 8                                      # pin0..7 = in(0)
    proof = SP.from_zkin(z)
    assert SP.stark_verify(proof, proof["rootC"], info, prog, ss, orc)
    for mutate in (lambda p: p["evals"][5].__setitem__(0, (p["evals"][5][0] + 1) % P),
                   lambda p: p["publics"].__setitem__(11, (p["publics"][11] + 1) % P),
                   lambda p: p["fri_proof"]["last"][3].__setitem__(1, p["fri_proof"]["last"][3][1] ^ 1),
                   lambda p: p["fri_proof"]["queries"][0]["pol_queries"][2][2][0].__setitem__(7, 5)):
        bad = copy.deepcopy(proof)
        mutate(bad)
        try:
            ok = SP.stark_verify(bad, bad["rootC"], info, prog, ss, orc)
        except ValueError as e:
            ok = "FRIVerifierFailed" not in str(e)
        assert not ok
    # a trace that breaks one round transition must not yield an accepted proof
    cm_bad = cm.copy(); cm_bad[19 * 12345 + 3] ^= 1
    ns = _native(stark, const, json.dumps(PG.program(nbits)), ss)
    zb = SP.from_zkin(ns.gen(cm_bad))
    ns.free()
    assert not SP.stark_verify(zb, zb["rootC"], info, prog, ss, orc)


@pytest.mark.parametrize("nbits", [12, 16])
def test_connection_pil_at_scale_zkin_equals_oracle(zk, orc, nbits):
    """grand product over 2^16 rows inside a proof (multi-block scan), cm3 = Z only, q_deg 1"""
    import stark_prover as SP, starkinfo as SI, poseidong as PG, connection_workload as CW
    stark = _stark(zk)
    ss = PG.stark_struct(nbits)
    const, cm = CW.make(nbits, orc.root(nbits), seed=nbits)
    su = SP.setup(CW.pil(nbits), const, ss, orc)
    exp = SP.to_zkin(SP.stark_gen(cm, su, ss, orc))
    ns = _native(stark, const, json.dumps(SI.to_json(su["starkinfo"], su["program"])), ss)
    got = ns.gen(cm)
    assert got == exp
    bad = cm.copy(); bad[3 * 1000 + 1] = (int(bad[3 * 1000 + 1]) + 1) % P     # breaks a copy constraint
    with pytest.raises(zk.ZkError, match="z does not close"):
        ns.gen(bad)
    ns.free()


@pytest.mark.parametrize("logn", [1, 9, 13, 16, 18, 20])
def test_calculate_z_matches_oracle(zk, orc, logn):
    """zk_stark_calculate_z_dev vs orc_calculate_z (stark_gen.rs:653-666) on random extension-field operands whose product
    closes; multi-block for logn > 12"""
    zk.init(0)
    n = 1 << logn
    rng = np.random.default_rng(logn)
    den = rng.integers(1, P, size=(n, 3), dtype=np.uint64)
    num = den[rng.permutation(n)]                                             # same multiset: the product of num/den is 1
    z_exp, ok = orc.calculate_z(num.reshape(-1), den.reshape(-1))
    assert ok
    d_num, d_den = zk.DevArray.from_host(num.reshape(-1)), zk.DevArray.from_host(den.reshape(-1))
    d_z = zk.DevArray(3 * n)
    zk._check(zk.lib().zk_stark_calculate_z_dev(d_num.ptr, d_den.ptr, n, d_z.ptr, None))
    assert np.array_equal(d_z.to_host(), z_exp)
    num2 = num.copy(); num2[n // 2, 1] ^= np.uint64(1)
    with pytest.raises(zk.ZkError, match="z does not close"):
        zk._check(zk.lib().zk_stark_calculate_z_dev(zk.DevArray.from_host(num2.reshape(-1)).ptr, d_den.ptr, n, d_z.ptr, None))


def _addp(a, b):
    """(a + b) mod p on canonical uint64 arrays"""
    s = a + b
    s = np.where(s < a, s + np.uint64(0xFFFFFFFF), s)                          # wrapped: 2^64 = 2^32 - 1 (mod p)
    return np.where(s >= np.uint64(P), s - np.uint64(P), s)


def test_lde_2p24_to_2p25_three_columns(zk, orc):
    """zk_gl_lde_dev at (24 -> 25, w = 3): transform plans the 2^10 fixtures never reach.  Checked by (i) the oracle's
    full-size extension of one of the three columns, (ii) additivity over the whole output, (iii) closed forms: a constant
    column extends to itself, the column X = (w_24^i) to (49 w_25^j), a zero column to zero."""
    zk.init(0)
    nbits, ext, w = 24, 25, 3
    n, nx = 1 << nbits, 1 << ext
    rng = np.random.default_rng(2425)
    a = rng.integers(0, P, size=n * w, dtype=np.uint64)
    b = rng.integers(0, P, size=n * w, dtype=np.uint64)
    lib = zk.lib()

    def lde(host):
        d_src, d_dst, d_tmp = zk.DevArray.from_host(host), zk.DevArray(nx * w), zk.DevArray(nx * w)
        zk._check(lib.zk_gl_lde_dev(d_src.ptr, w, nbits, d_dst.ptr, d_tmp.ptr, ext, None))
        return d_dst.to_host()
    ea, eb = lde(a), lde(b)
    col = np.ascontiguousarray(a.reshape(n, w)[:, 1])                           # (i)
    assert np.array_equal(ea.reshape(nx, w)[:, 1], orc.lde(col, 1, nbits, ext))
    assert np.array_equal(lde(_addp(a, b)), _addp(ea, eb))                      # (ii)
    s = np.zeros((n, w), np.uint64)                                             # (iii)
    s[:, 0] = 12345
    one = np.zeros(n, np.uint64); one[1] = 1
    s[:, 1] = orc.ntt(one, 1, nbits)                                            # the transform of delta_1 is (w_24^i)_i
    es = lde(s.reshape(-1)).reshape(nx, w)
    assert np.all(es[:, 0] == 12345) and not es[:, 2].any()
    one_x = np.zeros(nx, np.uint64); one_x[1] = 1
    wx = orc.ntt(one_x, 1, ext)                                                 # (w_25^j)_j
    c49 = np.full(nx, 49, np.uint64)
    acc = np.zeros(nx, np.uint64)
    for bit in range(6):                                                        # 49 * wx by double-and-add
        if (49 >> bit) & 1:
            acc = _addp(acc, wx)
        wx = _addp(wx, wx)
    assert np.array_equal(es[:, 1], acc)


def test_recursion_task_proofs_match_oracle(zk, orc):
    """BASELINE config 5's unit of work: the three STARKs of one recursion task (tools/aggregation_workload.py) through
    aggregation.ProverPool's path (tools/aggregation_workload.py pool) -- Fibonacci 2^10 and the compressor-shaped circuit at 2^15: zkin equal to the oracle prover's;
    at 2^18 (r1.starkStruct.json: 6 queries): accepted by the restated verifier."""
    import stark_prover as SP, starkinfo as SI, aggregation_workload as AW
    stark = _stark(zk)
    task = 5
    # 2^10 Fibonacci, the reference's own PIL and struct
    su = SP.setup(AW.fib_pil(), AW.fib_consts(), AW.STRUCTS["fib"], orc)
    exp = SP.to_zkin(SP.stark_gen(AW.fib_trace(task), su, AW.STRUCTS["fib"], orc))
    ns = _native(stark, AW.fib_consts(), json.dumps(AW.program("fib")), AW.STRUCTS["fib"])
    assert ns.gen(zk.DevArray.from_host(AW.fib_trace(task))) == exp
    ns.free()
    # 2^15 compressor-shaped circuit
    c = AW.Circuit(15)
    ss = AW.STRUCTS["c12"]
    su = SP.setup(AW.c12_pil(15), c.consts, ss, orc)
    assert AW.program("c12") == json.loads(json.dumps(SI.to_json(su["starkinfo"], su["program"])))
    cm = c.witness(task)
    exp = SP.to_zkin(SP.stark_gen(cm, su, ss, orc))
    ns = _native(stark, c.consts, json.dumps(AW.program("c12")), ss)
    got = ns.gen(zk.DevArray.from_host(cm))
    assert got == exp
    assert ns.gen(zk.DevArray.from_host(c.witness(task + 1)))["root1"] != got["root1"]   # another task, another witness
    ns.free()
    # 2^18: verifier only
    c = AW.Circuit(18)
    ss = AW.STRUCTS["r1"]
    info, prog, _ = SI.generate(AW.c12_pil(18), ss)
    ns = _native(stark, c.consts, json.dumps(AW.program("r1")), ss)
    p = SP.from_zkin(ns.gen(zk.DevArray.from_host(c.witness(task))))
    ns.free()
    assert len(p["fri_proof"]["queries"][0]["pol_queries"]) == 6
    assert SP.stark_verify(p, p["rootC"], info, prog, ss, orc)


def test_final_struct_bls12381_with_10_bit_fold(zk, orc):
    """the reference's final.starkStruct.bls12381.json (2^16 rows, FRI 17 -> 7 -> 3: ten bits in one fold, BLS12381 hashing) on
    the compressor-shaped circuit, scaled to 2^12 rows with the same 10-bit first fold: zkin equal to the oracle's"""
    import stark_prover as SP, starkinfo as SI, aggregation_workload as AW
    stark = _stark(zk)
    zk.bn128_init(field="bls12381")
    nbits = 12
    ss = {"nBits": nbits, "nBitsExt": nbits + 1, "nQueries": 8, "verificationHashType": "BLS12381", "steps": [{"nBits": 13}, {"nBits": 3}]}
    c = AW.Circuit(nbits)
    b = SP.BN128Backend(orc, "bls12381")
    su = SP.setup(AW.c12_pil(nbits), c.consts, ss, b)
    cm = c.witness(primary=list(range(1, 17)))
    exp = SP.to_zkin_bn128(SP.stark_gen(cm, su, ss, b), b, "addr")
    ns = stark.NativeStarkSetup(c.consts, json.dumps(SI.to_json(su["starkinfo"], su["program"])), json.dumps(ss), prover_addr="addr")
    got = ns.gen(cm)
    ns.free()
    assert got == exp
