"""The staged prover (include/zkgpu.h "the staged prover"): stark_gen cut at the reference's own seams -- calculate_exps_parallel
(stark_gen.rs:786-792), extend_and_merkelize (:709-750), the challenges, the evaluations, FRI::prove (fri.rs:84-184) -- must give, stage by
stage, the very proof zk_stark_gen gives in one call; and FRI::prove on its own (zk_fri_prove_dev), driven by a transcript the CALLER owns,
must give that proof's FRI part."""
import importlib
import json
import pathlib
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
D = ROOT / "tests" / "golden" / "starky_data"
CASES = {"fib_gl": ("fib.pil.json.gl", "fib.const.gl", "fib.cm.gl"), "plookup_gl": ("plookup.pil.json.gl", "plookup.const.gl", "plookup.cm.gl"),
         "fibonacci_imP": ("fib.pil.json", "fib.const", "fib.cm"), "permutation": ("pe.pil.json", "pe.const", "pe.cm"),
         "connection": ("connection.pil.json", "connection.const", "connection.cm")}


def _struct(hash_type="GL", ext=1):
    return {"nBits": 10, "nBitsExt": 10 + ext, "nQueries": 8, "verificationHashType": hash_type, "steps": [{"nBits": 10 + ext}, {"nBits": 7}, {"nBits": 3}]}


def _setup(zk, name, ss):
    zk.init(0)
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    pil_f, const_f, cm_f = CASES[name]
    program = stark.generate_program(open(D / pil_f).read(), json.dumps(ss))
    return stark, stark.NativeStarkSetup(np.fromfile(D / const_f, dtype="<u8"), program, json.dumps(ss), prover_addr="1" if ss["verificationHashType"] != "GL" else None), np.fromfile(D / cm_f, dtype="<u8")


@pytest.mark.parametrize("hash_type", ["GL", "BN128"])
@pytest.mark.parametrize("name", list(CASES))
def test_staged_proof_equals_the_one_call_proof(zk, name, hash_type):
    stark, ns, cm = _setup(zk, name, _struct(hash_type))
    want = ns.gen_bytes(cm)
    assert ns.staged(cm).run_all() == want                                    # host trace
    assert ns.staged(zk.DevArray.from_host(cm)).run_all() == want             # HBM-resident trace
    ns.free()


def test_staged_proof_of_poseidong_with_early_stage3(zk):
    """PoseidonG: stage 3 depends on no challenge, so its columns start on the side stream at the first commitment -- same proof"""
    import poseidong as PG
    zk.init(0)
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    nbits = 12
    ns = stark.NativeStarkSetup(PG.consts(nbits), json.dumps(PG.program(nbits)), json.dumps(PG.stark_struct(nbits)))
    d_cm = zk.DevArray.from_host(PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits))
    assert ns.staged(d_cm).run_all() == ns.gen_bytes(d_cm)
    ns.free()


def test_a_staged_proof_keeps_its_device_trace_alive(zk):
    """zk_stark_new borrows d_cm_pols until zk_stark_free (zkgpu.h).  The Python mirror must hold the DevArray it was given: a temporary's
    __del__ hands the block back to the size-keyed pool, and the next reservation of that very size -- made here on purpose, and overwritten
    -- would be the trace the later stages read (advisor finding, round 5)."""
    stark, ns, cm = _setup(zk, "fibonacci_imP", _struct())
    want = ns.gen_bytes(cm)
    p = ns.staged(zk.DevArray.from_host(cm))                                  # a temporary: only the StagedProof keeps it
    import gc; gc.collect()
    junk = [zk.DevArray(cm.size, zero=True) for _ in range(4)]                # same size as the trace: would reuse its block if it had been freed
    p.commit_stage(1)
    more = [zk.DevArray(cm.size, zero=True) for _ in range(4)]
    p.challenge(0); p.challenge(1)
    p.eval(stark.STEP_2PREV); p.calculate_h1h2()
    p.commit_stage(2); p.challenge(2); p.challenge(3)
    p.eval(stark.STEP_3PREV); p.calculate_z(); p.eval(stark.STEP_3)
    p.commit_stage(3); p.challenge(4)
    p.eval(stark.STEP_42NS)
    p.commit_stage(4); p.challenge(7)
    p.evals(); p.challenge(5); p.challenge(6)
    p.eval(stark.STEP_52NS)
    p.fri_prove()
    assert p.finish() == want
    p.free()
    assert p._cm is None                                                      # released with the context, not before
    del junk, more
    ns.free()


def test_two_live_contexts_on_one_setup(zk):
    """PoseidonG's setup runs stage 3 early on the SETUP's side stream: a second context on the same setup must not share that stream's
    events (advisor finding, round 5) -- it takes stage 3 in order; both proofs equal the one-call proof"""
    import poseidong as PG
    zk.init(0)
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    nbits = 12
    ns = stark.NativeStarkSetup(PG.consts(nbits), json.dumps(PG.program(nbits)), json.dumps(PG.stark_struct(nbits)))
    d_cm = zk.DevArray.from_host(PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits))
    want = ns.gen_bytes(d_cm)
    a, b = ns.staged(d_cm), ns.staged(d_cm)
    a.commit_stage(1); b.commit_stage(1)                                      # a's early stage 3 is in flight on the side stream while b commits
    assert b.run_all_from_stage1() == want
    assert a.run_all_from_stage1() == want
    a.free(); b.free()
    assert ns.staged(d_cm).run_all() == want                                  # the side stream is free again
    ns.free()


def test_stages_are_accepted_in_the_references_order_only(zk):
    stark, ns, cm = _setup(zk, "fibonacci_imP", _struct())
    p = ns.staged(cm)
    with pytest.raises(zk.ZkError, match="in order"):
        p.commit_stage(2)
    with pytest.raises(zk.ZkError, match="first commitment"):
        p.eval(stark.STEP_2PREV)
    with pytest.raises(zk.ZkError, match="commitment it is drawn after"):
        p.challenge(0)                                                        # u is drawn from a transcript that holds root 1
    p.commit_stage(1)
    with pytest.raises(zk.ZkError, match="its challenges come first"):
        p.eval(stark.STEP_2PREV)                                              # u, defVal neither drawn nor set: would run on zeros (advisor finding, round 5)
    with pytest.raises(zk.ZkError, match="commitment it is drawn after"):
        p.set_challenge(2, [1, 2, 3])                                         # gamma belongs behind root 2
    p.challenge(0); p.challenge(1)
    with pytest.raises(zk.ZkError, match="calculate_H1H2"):
        p.commit_stage(2)
    p.eval(stark.STEP_2PREV)
    with pytest.raises(zk.ZkError, match="runs once"):
        p.eval(stark.STEP_2PREV)
    with pytest.raises(zk.ZkError, match="FRI::prove"):
        p.finish()
    with pytest.raises(zk.ZkError, match="challenge index"):
        p.challenge(8)
    p.free()
    assert ns.gen_bytes(cm) == ns.staged(cm).run_all()                        # an abandoned context leaves the setup usable
    ns.free()


def test_a_caller_with_its_own_transcript_and_fri_prove_alone(zk):
    """What a Rust caller that keeps its own stark_gen.rs does: ITS TranscriptGL absorbs the publics and the roots commit_stage hands out and
    squeezes the challenges it then sets; the evaluations come back from zk_stark_evals; FRI::prove runs through zk_fri_prove_dev with that
    transcript, the context's f polynomial and the five trees.  Everything equals the one-call proof."""
    stark, ns, cm = _setup(zk, "plookup_gl", _struct())
    z = json.loads(ns.gen_bytes(cm))
    p = ns.staged(cm)
    tr = zk.TranscriptGL()
    tr.put([int(v) for v in z["publics"]])
    words = lambda d: [int(v) for v in d] if isinstance(d, list) else [int(d), 0, 0, 0]
    def commit(stage):
        r = p.commit_stage(stage)
        assert r == words(z["root%d" % stage])
        tr.put(r)
    def chal(i):
        v = [int(x) for x in tr.get_field()]
        p.set_challenge(i, v)
    commit(1); chal(0); chal(1)
    p.eval(stark.STEP_2PREV); p.calculate_h1h2()
    commit(2); chal(2); chal(3)
    p.eval(stark.STEP_3PREV); p.calculate_z(); p.eval(stark.STEP_3)
    commit(3); chal(4)
    p.eval(stark.STEP_42NS)
    commit(4); chal(7)
    n_ev = len(z["evals"])
    out = np.zeros(3 * n_ev, np.uint64)
    assert zk.lib().zk_stark_evals(p._h, zk._ptr(out), out.size) == n_ev
    assert [[str(v) for v in out[3 * i:3 * i + 3]] for i in range(n_ev)] == z["evals"]
    tr.put([int(v) for v in out])
    chal(5); chal(6)
    p.eval(stark.STEP_52NS)
    ss = _struct()
    fri = stark.fri_prove_dev(tr._h, p.fri_pol_dev(), ss["nBitsExt"], [s["nBits"] for s in ss["steps"]], ss["nQueries"], [p.tree(j) for j in range(1, 6)])
    for k in ("s1_root", "s1_vals", "s1_siblings", "s2_root", "s2_vals", "s2_siblings", "finalPol"):
        assert fri[k] == z[k], k
    for j, nm in enumerate(["1", "2", "3", "4", "C"]):
        assert fri["s0_vals%d" % (j + 1)] == z["s0_vals" + nm] and fri["s0_siblings%d" % (j + 1)] == z["s0_siblings" + nm]
    assert len(fri["ys"]) == ss["nQueries"]
    p.free(); ns.free()
