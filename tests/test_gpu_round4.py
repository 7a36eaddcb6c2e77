"""Round 4: shapes the GPU suite had never run, and full-size proofs pinned byte for byte.

  * blow-up 4 and 8 (`nBitsExt = nBits + 2 / + 3`): `Zi` of period 2^ext, `next` strides of 4 / 8 rows in the run-time
    compiled kernels, q_deg up to 4, FRI from nBits + 2 (starkinfo.rs:173-175 only demands nBitsExt == steps[0].nBits;
    stark_gen.rs:575-592, interpreter.rs:194-197) -- on the reference's fixtures and on PoseidonG at 2^12 rows;
  * PoseidonG at 2^16 rows with BN128 hashing -- the flavour starky/README.md:51-57 benchmarks;
  * BASELINE config 3 (2^20 rows, FRI steps 21/15/11/7/4) and the 2^24-row headline proof against goldens the CPU oracle
    produced offline (tools/gen_golden_full.py): roots, evaluations, last polynomial, openings and sha256 of the whole zkin.
"""
import importlib
import json
import pathlib
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tools"))
D = ROOT / "tests" / "golden" / "starky_data"
G = ROOT / "tests" / "golden"


def _stark(zk):
    zk.init(0)
    return importlib.import_module("eigen_zkvm_amd.stark")


def _struct(ext_bits, hash_type="GL"):
    e = 10 + ext_bits
    return {"nBits": 10, "nBitsExt": e, "nQueries": 8, "verificationHashType": hash_type,
            "steps": [{"nBits": e}, {"nBits": 7}, {"nBits": 3}]}


FIXTURES = {"fib_gl": ("fib.pil.json.gl", "fib.const.gl", "fib.cm.gl"), "plookup_gl": ("plookup.pil.json.gl", "plookup.const.gl", "plookup.cm.gl"),
            "permutation": ("pe.pil.json", "pe.const", "pe.cm"), "connection": ("connection.pil.json", "connection.const", "connection.cm")}


@pytest.mark.parametrize("ext_bits", [2, 3])
@pytest.mark.parametrize("name", list(FIXTURES))
def test_fixture_proofs_at_blowup_4_and_8(zk, orc, name, ext_bits):
    import stark_prover as SP, starkinfo as SI
    stark = _stark(zk)
    ss = _struct(ext_bits)
    pil_f, const_f, cm_f = FIXTURES[name]
    pil = json.load(open(D / pil_f))
    su = SP.setup(pil, D / const_f, ss, orc)
    proof = SP.stark_gen(D / cm_f, su, ss, orc)
    assert SP.stark_verify(proof, proof["rootC"], su["starkinfo"], su["program"], ss, orc)
    exp = SP.to_zkin(proof)
    program = json.loads(stark.generate_program(json.dumps(pil), json.dumps(ss)))     # the product's own code generator
    assert program == json.loads(json.dumps(SI.to_json(su["starkinfo"], su["program"])))
    ns = stark.NativeStarkSetup(np.fromfile(D / const_f, dtype="<u8"), json.dumps(program), json.dumps(ss))
    got = ns.gen(np.fromfile(D / cm_f, dtype="<u8"))
    assert ns.verify(got) is True                                             # the library's own stark_verify at this blow-up
    ns.free()
    assert list(got) == list(exp)
    for k in exp:
        assert got[k] == exp[k], k


@pytest.mark.parametrize("hash_type", ["GL", "BN128"])
def test_plookup_blowup_4_scalar_field_hash(zk, orc, hash_type):
    """the reference's own BN128 fixtures at blow-up 4 (16-ary trees over 2^12 rows)"""
    import stark_prover as SP, starkinfo as SI
    stark = _stark(zk)
    ss = _struct(2, hash_type)
    b = orc if hash_type == "GL" else SP.BN128Backend(orc)
    pil = json.load(open(D / "plookup.pil.json"))
    su = SP.setup(pil, D / "plookup.const", ss, b)
    proof = SP.stark_gen(D / "plookup.cm", su, ss, b)
    assert SP.stark_verify(proof, proof["rootC"], su["starkinfo"], su["program"], ss, b)
    exp = SP.to_zkin(proof) if hash_type == "GL" else SP.to_zkin_bn128(proof, b, "7")
    ns = stark.NativeStarkSetup(np.fromfile(D / "plookup.const", dtype="<u8"), json.dumps(SI.to_json(su["starkinfo"], su["program"])),
                                json.dumps(ss), prover_addr="7")
    got = ns.gen(np.fromfile(D / "plookup.cm", dtype="<u8"))
    assert ns.verify(got) is True
    ns.free()
    for k in exp:
        assert got[k] == exp[k], k


@pytest.mark.parametrize("nbits,ext_bits", [(12, 2), (11, 3)])
def test_poseidong_blowup_4_zkin_equals_oracle(zk, orc, nbits, ext_bits):
    """PoseidonG with more room for the constraint degree: fewer intermediate columns (24 at blow-up 4), q_deg 4"""
    import stark_prover as SP, starkinfo as SI, poseidong as PG
    stark = _stark(zk)
    ss, const = PG.stark_struct(nbits, ext_bits=ext_bits), PG.consts(nbits)
    cm = PG.trace(nbits, None, PG.FIRST_COUNT, seed=nbits)
    su = SP.setup(PG.pil(nbits), const, ss, orc)
    if ext_bits == 2:
        assert (su["starkinfo"]["n_cm3"], su["starkinfo"]["q_deg"]) == (24, 4)
    proof = SP.stark_gen(cm, su, ss, orc)
    assert SP.stark_verify(proof, proof["rootC"], su["starkinfo"], su["program"], ss, orc)
    exp = SP.to_zkin(proof)
    program = PG.program(nbits, ss)
    assert program == json.loads(json.dumps(SI.to_json(su["starkinfo"], su["program"])))
    ns = stark.NativeStarkSetup(const, json.dumps(program), json.dumps(ss))
    got = ns.gen(zk.DevArray.from_host(cm))
    assert list(got) == list(exp)
    for k in exp:
        assert got[k] == exp[k], k
    assert ns.gen(cm) == got
    assert ns.verify(got) is True
    ns.free()


def test_poseidong_2p16_bn128_hash_zkin_equals_oracle(zk, orc):
    """MerkleTreeBN128 + TranscriptBN128 over the PoseidonG sections (19 / 36 / 6 / 18 words per row, 2^17 rows)"""
    import stark_prover as SP, starkinfo as SI, poseidong as PG
    stark = _stark(zk)
    nbits = 16
    ss, const = PG.stark_struct(nbits, hash_type="BN128"), PG.consts(nbits)
    cm = PG.trace(nbits, None, PG.FIRST_COUNT, seed=nbits)
    b = SP.BN128Backend(orc)
    su = SP.setup(PG.pil(nbits), const, ss, b)
    proof = SP.stark_gen(cm, su, ss, b)
    assert SP.stark_verify(proof, proof["rootC"], su["starkinfo"], su["program"], ss, b)
    exp = SP.to_zkin_bn128(proof, b, "")
    ns = stark.NativeStarkSetup(const, json.dumps(PG.program(nbits, ss)), json.dumps(ss))
    got = ns.gen(zk.DevArray.from_host(cm))
    assert ns.verify(got) is True                                             # 16-ary trees, scalar-field sponge
    bad = json.loads(json.dumps(got)); bad["s0_vals3"][1][5] = str((int(bad["s0_vals3"][1][5]) + 1) % 0xFFFFFFFF00000001)
    assert ns.verify(bad) is False
    ns.free()
    assert list(got) == list(exp)
    for k in exp:
        assert got[k] == exp[k], k


def _check_against_golden(got, gold):
    from gen_golden_full import zkin_digest
    for k in gold:
        if k in got:
            assert got[k] == gold[k], k                                       # roots, evals, publics, finalPol, s{i}_root
    for k, d in gold.get("openings_digest", {}).items():
        assert zkin_digest(got[k]) == d, k
    if "zkin_digest" in gold:
        assert zkin_digest(got) == gold["zkin_digest"]


def _prove_poseidong(zk, nbits, gold):
    import poseidong as PG
    stark = _stark(zk)
    ss = PG.stark_struct(nbits)
    assert ss == gold["starkStruct"]
    const = PG.consts(nbits)
    ns = stark.NativeStarkSetup(const, json.dumps(PG.program(nbits, ss)), json.dumps(ss))
    del const
    assert [str(v) for v in ns.const_root()] == gold["rootC"]
    cm = PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits)
    d_cm = zk.DevArray.from_host(cm)
    del cm
    got = ns.gen(d_cm)
    assert ns.verify(got) is True                                             # zk_stark_verify at full size
    ns.free()
    return got


def test_config3_2p20_proof_equals_offline_oracle_golden(zk):
    """BASELINE config 3 with the stated struct (nBits 20, steps 21/15/11/7/4): the whole zkin, byte for byte"""
    gold = json.load(open(G / "poseidong_2p20.json"))
    assert [s["nBits"] for s in gold["starkStruct"]["steps"]] == [21, 15, 11, 7, 4]
    got = _prove_poseidong(zk, 20, gold)
    _check_against_golden(got, gold)
    assert "zkin_digest" in gold


def test_headline_2p24_proof_equals_offline_oracle_golden(zk):
    """the 2^24-row headline proof bench.py times (nBitsExt 25, steps 25/20/15/10/5): rootC and root1 from the oracle's
    LDE + Merkle, and -- when the build container had the memory for the whole oracle proof -- everything else"""
    f = G / "poseidong_2p24.json"
    gold = json.load(open(f if f.exists() else G / "poseidong_2p24_roots.json"))
    assert [s["nBits"] for s in gold["starkStruct"]["steps"]] == [25, 20, 15, 10, 5]
    got = _prove_poseidong(zk, 24, gold)
    _check_against_golden(got, gold)


def test_random_starkstructs_on_fixtures_and_wide_fibonacci():
    """a seeded slice of the round-4 fuzz campaign (tools/fuzz_proofs.py; profiles/r04/fuzz.txt): the reference's fixtures and the
    wide-Fibonacci PIL under random StarkStructs -- blow-up 2 / 4 / 8, 1..16 queries, random FRI steps (folds of 1..8 bits), the three hash
    types -- device zkin == oracle zkin, accepted by zk_stark_verify and by the oracle's verifier, a tampered copy rejected"""
    import fuzz_proofs
    assert fuzz_proofs.run(404, 24, verbose=False) == []


def test_random_shapes_of_the_primitives_and_random_tamperings():
    """seeded slices of the other two fuzzers (tools/fuzz_primitives.py, tools/fuzz_verify.py): NTT / LDE / Merkle (GL and both scalar
    fields, any height) / FRI folds / G1 sums at random shapes against the oracle; random single-word tamperings of GL, BN128 and
    BLS12381 proofs -- the library's verdict equals the oracle's"""
    import fuzz_primitives, fuzz_verify
    assert fuzz_primitives.run(77, 40, verbose=False) == []
    assert fuzz_verify.run(77, 25, verbose=False) == []
