"""The full-size goldens (tests/golden/poseidong_2p20.json, poseidong_2p24.json; made offline by tools/gen_golden_full.py from the
oracle prover alone): well-formed, for the stated configurations, and the generator's summary / digest functions reproduce on a size
the CPU suite can afford -- so that the GPU tests that compare against them (tests/test_gpu_round4.py) compare like with like."""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tools"))
G = ROOT / "tests" / "golden"


def test_committed_goldens_are_for_the_stated_configs():
    import poseidong as PG
    for nbits, steps in ((20, [21, 15, 11, 7, 4]), (24, [25, 20, 15, 10, 5])):          # config 3 mirrors r2.starkStruct.bn128.json; the headline
        g = json.load(open(G / ("poseidong_2p%d.json" % nbits)))
        assert g["nBits"] == nbits and g["starkStruct"] == PG.stark_struct(nbits)
        assert [s["nBits"] for s in g["starkStruct"]["steps"]] == steps and g["starkStruct"]["nBitsExt"] == nbits + 1
        assert len(g["zkin_digest"]) == 64 and all(len(d) == 64 for d in g["openings_digest"].values())
        assert len(g["evals"]) == 91 and all(len(e) == 3 for e in g["evals"])               # PoseidonG's ev_map
        assert len(g["finalPol"]) == 1 << steps[-1] and len(g["publics"]) == 12
        assert sorted(k for k in g if k.endswith("_root")) == ["s%d_root" % i for i in range(1, len(steps))]
        assert set(g["openings_digest"]) == {"s0_vals%s" % t for t in "1234C"} | {"s0_siblings%s" % t for t in "1234C"} | \
            {"s%d_%s" % (i, k) for i in range(1, len(steps)) for k in ("vals", "siblings")}
        assert "gen_golden_full.py" in g["generator"]


def test_generator_summary_reproduces_at_small_size(orc):
    import gen_golden_full as GG, poseidong as PG, stark_prover as SP
    nbits = 10
    ss = PG.stark_struct(nbits)
    su = SP.setup(PG.pil(nbits), PG.consts(nbits), ss, orc)
    cm = PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits)
    z = SP.to_zkin(SP.stark_gen(cm, su, ss, orc))
    z_lean = SP.to_zkin(SP.stark_gen(cm, su, ss, orc, lean=True))                          # the memory-lean path the 2^24 golden used
    assert z == z_lean
    s = GG.summary(z, nbits, ss, 0.0)
    assert s["zkin_digest"] == GG.zkin_digest(json.loads(json.dumps(z)))                   # the digest survives a JSON round trip (what the GPU tests hash)
    assert s["rootC"] == z["rootC"] and s["evals"] == z["evals"] and s["openings_digest"]["s0_vals1"] == GG.zkin_digest(z["s0_vals1"])
    proof = SP.from_zkin(z)
    assert SP.stark_verify(proof, proof["rootC"], su["starkinfo"], su["program"], ss, orc)
