"""Parity of compressor12 exec on the device (csrc/compressor12.hip, through the C ABI) against oracle/compressor12.py:
bit exact.  SURVEY.md 8(f)-4."""
import importlib, pathlib, random, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import compressor12 as C12  # noqa: E402
P = C12.P


@pytest.fixture(scope="module")
def dev(zk):
    assert zk.lib().zk_device_count() >= 1, "no GPU visible (the product has no CPU fallback)"
    zk.init(0)
    return importlib.import_module("eigen_zkvm_amd.compressor12")


def make_case(rng, n_wit, n_adds, n_map, chain):
    """adds reading random earlier wires; `chain`: every add also reads its predecessor (depth = n_adds)"""
    adds = []
    for i in range(n_adds):
        hi = n_wit + i
        a = hi - 1 if chain and i else rng.randrange(hi)
        adds.append((a, rng.randrange(hi), rng.choice([1, P - 1, rng.randrange(P)]), rng.randrange(P)))
    s_map = [[rng.choice([0, rng.randrange(n_wit + n_adds)]) for _ in range(n_map)] for _ in range(12)]
    wit = [1] + [rng.choice([0, P - 1, rng.randrange(P)]) for _ in range(n_wit - 1)]
    return C12.write_exec(adds, s_map), wit


@pytest.mark.parametrize("n_wit,n_adds,n_map,n_rows,chain", [(5, 0, 3, 4, False), (40, 25, 16, 16, False), (300, 500, 1000, 1024, False),
                                                             (50, 200, 64, 64, True), (3000, 20000, 4000, 4096, False)])
def test_exec_matches_oracle(zk, dev, n_wit, n_adds, n_map, n_rows, chain):
    rng = random.Random(n_wit * 7 + n_adds)
    text, wit = make_case(rng, n_wit, n_adds, n_map, chain)
    E = dev.Compressor12Exec(text, n_wit)
    assert E.depth == (n_adds if chain else E.depth) and (n_adds == 0) == (E.depth == 0)
    cm = E.run(np.array(wit, dtype=np.uint64), n_rows).to_host().reshape(n_rows, 12)
    assert np.array_equal(cm, C12.exec_cm(text, wit, n_rows))
    E.free()


def test_exec_errors(zk, dev):
    rng = random.Random(1)
    text, wit = make_case(rng, 10, 5, 4, False)
    with pytest.raises(zk.ZkError, match="length does not match"):
        dev.Compressor12Exec(text[:-1] + ",7]", 10)
    with pytest.raises(zk.ZkError, match="array of integers"):
        dev.Compressor12Exec("{}", 10)
    with pytest.raises(zk.ZkError, match="does not exist yet"):
        dev.Compressor12Exec(C12.write_exec([(12, 0, 1, 1)], [[0]] * 12), 10)
    E = dev.Compressor12Exec(text, 10)
    with pytest.raises(zk.ZkError, match="witness has"):
        E.run(np.array(wit[:-1], dtype=np.uint64), 4)
    with pytest.raises(zk.ZkError, match="more rows than the trace"):
        E.run(np.array(wit, dtype=np.uint64), 2)
    bad = list(wit); bad[3] = P
    with pytest.raises(zk.ZkError, match="not a Goldilocks field element"):
        E.run(np.array(bad, dtype=np.uint64), 4)
    E.free()


@pytest.mark.parametrize("nbits,layer_bits", [(8, 4), (12, 8), (16, 12)])
def test_join_circuit_exec_matches_oracle_and_host_walk(zk, dev, nbits, layer_bits):
    """The aggregation's join circuit (tools/aggregation_workload.py JoinCircuit: the recursive2 stand-in of
    test/stark_aggregation.sh:80-156 whose witness compressor12 exec can compute): the device exec of its .exec file gives the
    oracle's .cm matrix (oracle/compressor12.py, up to 2^12 rows: it is a Python loop) and the trace the host walk over the
    gates gives (tools/tracegen.c c12s_witness), which satisfies every identity of the PIL (the proof below is accepted)."""
    sys.path.insert(0, str(ROOT / "tools"))
    import aggregation_workload as AW
    J = AW.JoinCircuit(nbits, layer_bits)
    text = J.exec_text()
    E = dev.Compressor12Exec(text, AW.JoinCircuit.N_WITNESS)
    assert 1 <= E.depth <= (1 << nbits) >> layer_bits                 # at most one level of additions per layer
    for seed in (0, 1):
        primary = np.random.default_rng(seed).integers(0, P, 16, dtype=np.uint64)
        if seed == 0: primary[:3] = [0, P - 1, 1]
        w = J.witness_vector(primary)
        cm = E.run(w, 1 << nbits).to_host()
        assert np.array_equal(cm, J.witness(primary))
        if nbits <= 12:
            assert np.array_equal(cm.reshape(-1, 12), C12.exec_cm(text.decode(), [int(x) for x in w], 1 << nbits))
    E.free()


def test_join_proof_from_device_exec_verifies(zk, dev, orc):
    """exec on the device -> zk_stark_gen_dev on the trace it left in HBM -> the restated verifier accepts, the publics are
    the first three primary inputs, and the proof equals the one made from the host-walk trace"""
    import json
    sys.path.insert(0, str(ROOT / "tools"))
    import aggregation_workload as AW
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    import stark_prover as SP, starkinfo as SI                         # oracle/: the checker
    nbits = 10
    ss = {"nBits": nbits, "nBitsExt": nbits + 1, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": nbits + 1}, {"nBits": 7}, {"nBits": 3}]}
    J = AW.JoinCircuit(nbits, 6)
    import poseidong
    prog = poseidong.native_program(AW.c12_pil(nbits), ss)
    S = stark.NativeStarkSetup(J.consts, json.dumps(prog), json.dumps(ss))
    E = dev.Compressor12Exec(J.exec_text(), AW.JoinCircuit.N_WITNESS)
    primary = np.arange(100, 116, dtype=np.uint64)
    d_cm = E.run(J.witness_vector(primary), 1 << nbits)
    proof = S.gen(d_cm)
    assert [int(x) for x in proof["publics"]] == [100, 101, 102]
    assert proof == S.gen(J.witness(primary))
    p = SP.from_zkin(proof)
    assert [int(v) for v in p["rootC"]] == S.const_root()
    info, oprog, _ = SI.generate(AW.c12_pil(nbits), ss)
    assert SP.stark_verify(p, p["rootC"], info, oprog, ss, orc)
    E.free(); S.free()
