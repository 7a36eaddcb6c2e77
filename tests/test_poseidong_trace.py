"""The PoseidonG inputs of BASELINE config 3 (tools/tracegen.c, semantics of starkjs/poseidon/sm_poseidong.js): the generated
trace carries the reference's Poseidon known answers, satisfies every identity of the compiled PIL row by row, and the oracle
prover/verifier accept it; the C and the pure-Python interpreters of the oracle agree."""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tools"))
P = 0xFFFFFFFF00000001


def test_trace_rows_carry_the_poseidon_known_answers(orc, golden):
    import poseidong as PG
    for first, key in ((PG.FIRST_ZERO, 0), (PG.FIRST_COUNT, 1)):
        t = PG.trace(10, 1, first).reshape(-1, 19)
        exp = [int(v) for v in orc.poseidon(np.array(first[:8], np.uint64), np.array(first[8:], np.uint64), 4)]
        assert [int(v) for v in t[30, :4]] == exp and [int(v) for v in t[5, 12:16]] == exp
    z = PG.trace(10, 1, PG.FIRST_ZERO).reshape(-1, 19)
    assert [int(v) for v in z[30, :4]] == [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca]  # poseidon_opt.rs:219-230
    full = PG.trace(12, None, PG.FIRST_COUNT, seed=5).reshape(-1, 19)
    for blk in (1, 17, (1 << 12) // 31 - 1):                              # every slot is a real permutation of its own input
        inp = full[31 * blk, :12]
        assert np.array_equal(full[31 * blk + 30, :4], orc.poseidon(inp[:8], inp[8:], 4))
    assert np.array_equal(full[-1, 12:16], z[30, :4])                     # the partial last block is zero-input padding
    try:
        PG.trace(10, 34)
        assert False
    except ValueError as e:
        assert "Not enough Poseidon slots" in str(e)                       # sm_poseidong.js:151-153


def test_every_identity_holds_on_every_row():
    """evaluate the compiled PIL's identities directly (Python ints), all 2^10 rows"""
    import poseidong as PG
    nbits = 10; N = 1 << nbits
    pil = PG.pil(nbits)
    cm = PG.trace(nbits, None, PG.FIRST_COUNT, seed=1).reshape(N, 19).astype(object)
    cn = PG.consts(nbits).reshape(N, 18).astype(object)
    pub = [int(cm[p["idx"], p["polId"]]) for p in pil["publics"]]
    ex = pil["expressions"]

    def ev(e, i, memo):
        op = e["op"]
        if op == "number": return int(e["value"]) % P
        if op == "cm": return cm[(i + (1 if e.get("next") else 0)) % N, e["id"]]
        if op == "const": return cn[(i + (1 if e.get("next") else 0)) % N, e["id"]]
        if op == "public": return pub[e["id"]]
        if op == "exp":
            k = (e["id"], bool(e.get("next")))
            if k not in memo:
                memo[k] = ev(ex[e["id"]], (i + (1 if e.get("next") else 0)) % N, {})
            return memo[k]
        a, b = (ev(v, i, memo) for v in e["values"])
        return {"add": a + b, "sub": a - b, "mul": a * b}[op] % P
    for i in list(range(70)) + list(range(N - 40, N)):
        memo = {}
        for pi in pil["polIdentities"]:
            assert ev(ex[pi["e"]], i, memo) == 0, (i, pi)


def test_oracle_proves_and_verifies_poseidong_and_interpreters_agree(orc):
    import stark_prover as SP, poseidong as PG
    nbits = 10
    ss = PG.stark_struct(nbits)
    su = SP.setup(PG.pil(nbits), PG.consts(nbits), ss, orc)
    info = su["starkinfo"]
    assert (info["n_cm3"], info["q_deg"], info["q_dim"], info["map_sectionsN"]["cm4_2ns"]) == (36, 2, 3, 6)
    cm = PG.trace(nbits, None, PG.FIRST_COUNT, seed=3)
    proof = SP.stark_gen(cm, su, ss, orc)
    assert SP.stark_verify(proof, proof["rootC"], info, su["program"], ss, orc)
    assert SP.stark_gen(cm, su, ss, orc, use_c=False) == proof            # oracle/interp.c == oracle/interp.py
    back = SP.from_zkin(json.loads(json.dumps(SP.to_zkin(proof))))
    assert SP.stark_verify(back, back["rootC"], info, su["program"], ss, orc)
    bad = cm.copy(); bad[19 * 40 + 3] ^= np.uint64(1)                     # one wrong state word
    pb = SP.stark_gen(bad, su, ss, orc)
    assert not SP.stark_verify(pb, pb["rootC"], info, su["program"], ss, orc)


def test_c_interpreter_equals_python_on_reference_fixtures(orc):
    import stark_prover as SP
    D = ROOT / "tests" / "golden" / "starky_data"
    GL = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}
    for pil_f, c, m in (("plookup.pil.json.gl", "plookup.const.gl", "plookup.cm.gl"), ("connection.pil.json", "connection.const", "connection.cm"),
                        ("fib.pil.json", "fib.const", "fib.cm")):
        su = SP.setup(json.load(open(D / pil_f)), D / c, GL, orc)
        assert SP.stark_gen(D / m, su, GL, orc, use_c=True) == SP.stark_gen(D / m, su, GL, orc, use_c=False)


def test_connection_workload_closes(orc):
    import stark_prover as SP, poseidong as PG, connection_workload as CW
    nbits = 11
    ss = PG.stark_struct(nbits)
    const, cm = CW.make(nbits, orc.root(nbits), seed=2)
    su = SP.setup(CW.pil(nbits), const, ss, orc)
    proof = SP.stark_gen(cm, su, ss, orc)
    assert SP.stark_verify(proof, proof["rootC"], su["starkinfo"], su["program"], ss, orc)
