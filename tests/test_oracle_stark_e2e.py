"""End-to-end checks of the oracle's restated prover (codegen + stark_gen + FRI) and verifier on the
reference's own fixtures: the reference pins full proofs only by "the verifier accepts"
(stark_gen.rs:981-1195, serializer.rs:601-648), so that is the bar here too."""
import copy
import json
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
D = ROOT / "tests" / "golden" / "starky_data"
GL_STRUCT = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL",
             "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}

CASES = {
    # reference test                      pil                    const               cm
    "fib_gl (serializer.rs:601-648)":     ("fib.pil.json.gl",     "fib.const.gl",     "fib.cm.gl"),
    "plookup_gl (stark_gen.rs:1148)":     ("plookup.pil.json.gl", "plookup.const.gl", "plookup.cm.gl"),
    "fibonacci (stark_gen.rs:981)":       ("fib.pil.json",        "fib.const",        "fib.cm"),
    "permutation (stark_gen.rs:1030)":    ("pe.pil.json",         "pe.const",         "pe.cm"),
    "connection (stark_gen.rs:1101)":     ("connection.pil.json", "connection.const", "connection.cm"),
}


def _rejects(SP, proof, root, info, prog, orc):
    """the reference verifier answers Ok(false) or bails (Err) on a bad proof (stark_verify.rs:86-101)"""
    try:
        return not SP.stark_verify(proof, root, info, prog, GL_STRUCT, orc)
    except ValueError as e:
        return "FRIVerifierFailed" in str(e)


@pytest.mark.parametrize("name", list(CASES))
def test_prove_then_verify_accepts(orc, name):
    import stark_prover as SP
    pil_f, const_f, cm_f = CASES[name]
    pil = json.load(open(D / pil_f))
    su = SP.setup(pil, D / const_f, GL_STRUCT, orc)
    proof = SP.stark_gen(D / cm_f, su, GL_STRUCT, orc)
    info, prog = su["starkinfo"], su["program"]
    assert SP.stark_verify(proof, proof["rootC"], info, prog, GL_STRUCT, orc)
    bad = copy.deepcopy(proof)                                  # a single flipped opening must be rejected
    bad["evals"][0][0] = (bad["evals"][0][0] + 1) % SP.P
    assert _rejects(SP, bad, proof["rootC"], info, prog, orc)
    bad = copy.deepcopy(proof)
    bad["fri_proof"]["last"][1][2] ^= 1
    assert _rejects(SP, bad, proof["rootC"], info, prog, orc)
    z = SP.to_zkin(proof)                                        # serializer.rs:146-261 key set and order
    n_steps = len(GL_STRUCT["steps"])
    exp = ["rootC", "root1", "root2", "root3", "root4", "evals"]
    for i in range(1, n_steps):
        exp += ["s%d_root" % i, "s%d_vals" % i, "s%d_siblings" % i]
    exp += ["s0_vals1", "s0_vals2", "s0_vals3", "s0_vals4", "s0_valsC", "s0_siblings1", "s0_siblings2", "s0_siblings3",
            "s0_siblings4", "s0_siblingsC", "finalPol", "publics"]
    assert list(z) == exp
    json.dumps(z)


def test_fib_gl_const_root_is_the_reference_kat(orc, golden):
    import stark_prover as SP
    su = SP.setup(json.load(open(D / "fib.pil.json.gl")), D / "fib.const.gl", GL_STRUCT, orc)
    assert [int(v) for v in su["const_tree"][-4:]] == golden["const_root_fib_gl"]["root"]   # stark_setup.rs:100-116


# ---- verificationHashType "BN128": the instantiation the reference's own tests prove (stark_gen.rs:981-1146) ----
BN128_STRUCT = dict(GL_STRUCT, verificationHashType="BN128")                 # starky/data/starkStruct.json


def test_bn128_const_root_known_answer(orc):
    """stark_setup.rs:83-97: LDE + MerkleTreeBN128 of data/fib.const"""
    import stark_prover as SP
    b = SP.BN128Backend(orc)
    su = SP.setup(json.load(open(D / "fib.pil.json")), D / "fib.const", BN128_STRUCT, b)
    assert b.digest_str(su["const_tree"][-4:]) == "4658128321472362347225942316135505030498162093259225938328465623672244875764"


@pytest.mark.parametrize("pil_f,const_f,cm_f", [("fib.pil.json", "fib.const", "fib.cm"), ("plookup.pil.json", "plookup.const", "plookup.cm")])
def test_bn128_prove_then_verify_and_tamper(orc, pil_f, const_f, cm_f):
    import stark_prover as SP
    b = SP.BN128Backend(orc)
    su = SP.setup(json.load(open(D / pil_f)), D / const_f, BN128_STRUCT, b)
    proof = SP.stark_gen(D / cm_f, su, BN128_STRUCT, b)
    assert SP.stark_verify(proof, proof["rootC"], su["starkinfo"], su["program"], BN128_STRUCT, b)
    bad = copy.deepcopy(proof)
    bad["evals"][0][0] = (bad["evals"][0][0] + 1) % 0xFFFFFFFF00000001
    try:
        ok = SP.stark_verify(bad, bad["rootC"], su["starkinfo"], su["program"], BN128_STRUCT, b)
    except ValueError:
        ok = False
    assert not ok
    z = SP.to_zkin_bn128(proof, b, "addr")
    assert list(z)[-1] == "proverAddr" and len(z["s0_siblings1"][0][0]) == 16
    # the zkin text alone is enough for the verifier (bench.py checks the aggregation's final STARK this way)
    back = SP.from_zkin_bn128(json.loads(json.dumps(z)), b)
    assert SP.stark_verify(back, back["rootC"], su["starkinfo"], su["program"], BN128_STRUCT, b)
    z2 = json.loads(json.dumps(z)); 
    # (merklehash_bn128.rs:108-128 re-hashes each level's group and never splices the running value in -- `cur_idx` is commented
    # out there -- so only the LAST group of a path binds the root; the restatement keeps that behaviour, and so does this test)
    z2["s0_siblings1"][1][-1][5] = str(int(z2["s0_siblings1"][1][-1][5]) + 1)
    try:
        ok = SP.stark_verify(SP.from_zkin_bn128(z2, b), back["rootC"], su["starkinfo"], su["program"], BN128_STRUCT, b)
    except ValueError:
        ok = False
    assert not ok
