"""The product's code generator (csrc/starkinfo_gen.hip, zk_starkinfo_generate: StarkInfo::new, starkinfo.rs:160-272) against the
oracle's restatement (oracle/starkinfo.py) on every PIL in the tree: the two were written separately, in different languages,
from the same reference sources; their JSON must be equal value for value -- including the order of ev_map / ev_idx, the
numbering of temporaries and the section positions, which all end up in the proof.  Host only (no GPU)."""
import importlib
import json
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tools"))
D = ROOT / "tests" / "golden" / "starky_data"
GL = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}


def _both(zk, pil, ss):
    import starkinfo as SI
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    info, prog, _ = SI.generate(pil, ss)
    exp = json.loads(json.dumps(SI.to_json(info, prog)))
    got = json.loads(stark.generate_program(json.dumps(pil), json.dumps(ss)))
    return got, exp


@pytest.mark.parametrize("name", ["fib.pil.json", "fib.pil.json.gl", "plookup.pil.json", "plookup.pil.json.gl", "pe.pil.json", "connection.pil.json"])
def test_reference_fixtures(zk, name):
    got, exp = _both(zk, json.load(open(D / name)), GL)
    for part in ("starkinfo", "program"):
        for k in exp[part]:
            assert got[part][k] == exp[part][k], (part, k)
    assert got == exp


def test_blowup_4_and_bn128_struct(zk):
    """another extension factor changes the intermediate-polynomial search (starkinfo_cp_prover.rs:36-48)"""
    ss = {"nBits": 10, "nBitsExt": 12, "nQueries": 4, "verificationHashType": "BN128", "steps": [{"nBits": 12}, {"nBits": 7}, {"nBits": 3}]}
    for name in ("fib.pil.json", "plookup.pil.json", "connection.pil.json"):
        got, exp = _both(zk, json.load(open(D / name)), ss)
        assert got == exp, name


def test_poseidong_and_compressor_shape(zk):
    import poseidong as PG, aggregation_workload as AW
    for pil, ss in ((PG.pil(10), PG.stark_struct(10)), (PG.pil(16), PG.stark_struct(16)), (AW.c12_pil(15), AW.STRUCTS["c12"]),
                    (PG.pil(12), dict(PG.stark_struct(12), nBitsExt=14, steps=[{"nBits": 14}, {"nBits": 9}, {"nBits": 4}]))):
        got, exp = _both(zk, pil, ss)
        assert got == exp
    got, _ = _both(zk, PG.pil(10), PG.stark_struct(10))
    assert (got["starkinfo"]["n_cm3"], got["starkinfo"]["q_deg"]) == (36, 2)


def test_errors(zk):
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    pil = json.load(open(D / "fib.pil.json"))
    with pytest.raises(zk.ZkError, match="stark_deg != pil_deg"):
        stark.generate_program(json.dumps(pil), json.dumps(dict(GL, nBits=11, nBitsExt=12, steps=[{"nBits": 12}])))
    with pytest.raises(zk.ZkError, match="MustEqualDegreeError"):
        stark.generate_program(json.dumps(pil), json.dumps(dict(GL, steps=[{"nBits": 10}])))
    conn = json.load(open(D / "connection.pil.json"))
    del conn["references"]["Global.L1"]
    with pytest.raises(zk.ZkError, match="Global.L1 must be defined"):      # stark_setup.rs:25-31, starkinfo_Z.rs
        stark.generate_program(json.dumps(conn), json.dumps(GL))
    with pytest.raises(zk.ZkError, match="json"):
        stark.generate_program("{", json.dumps(GL))
