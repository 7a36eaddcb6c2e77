"""Proofs of different setups running at the same time, each from its own host thread on its own non-blocking stream
(zk_stark_gen_dev_on): every proof must be the proof the same setup gives alone on the default stream.  This is how the
recursion tasks of BASELINE config 5 share one GPU (test/stark_aggregation.sh:70-73 runs them as parallel processes), and it
exercises the cross-stream ordering of the caching allocator (csrc/capi.hip)."""
import json
import pathlib
import sys
import threading

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


@pytest.mark.gpu
def test_concurrent_proofs_equal_sequential_proofs(zk):
    import importlib
    import aggregation_workload as AW
    stark = importlib.import_module("eigen_zkvm_amd.stark")
    assert zk.lib().zk_device_count() >= 1
    zk.init(0)
    import poseidong as PG
    W, ROUNDS = 6, 3
    circ = AW.Circuit(AW.STRUCTS["c12"]["nBits"])
    consts = {"fib": AW.fib_consts(), "c12": circ.consts, "pg": PG.consts(10)}
    programs = {"fib": AW.program("fib"), "c12": AW.program("c12"), "pg": PG.program(10)}   # pg: stage 3 runs early on the setup's side stream
    structs = {"fib": AW.STRUCTS["fib"], "c12": AW.STRUCTS["c12"], "pg": PG.stark_struct(10)}
    trace = {"fib": lambda k: AW.fib_trace(k), "c12": lambda k: circ.witness(k), "pg": lambda k: PG.trace(10, None, PG.FIRST_COUNT, seed=k)}
    workers = []
    for w in range(W):
        kind = ("fib", "c12", "pg")[w % 3]
        su = stark.NativeStarkSetup(consts[kind], json.dumps(programs[kind]), json.dumps(structs[kind]))
        cms = [zk.DevArray.from_host(trace[kind](10 * w + r)) for r in range(ROUNDS)]
        workers.append((su, cms))
    alone = [[su.gen(cm) for cm in cms] for su, cms in workers]            # default stream, one at a time
    assert len({json.dumps(z, sort_keys=True) for zs in alone for z in zs}) >= W * ROUNDS - 2   # (the Fibonacci inputs repeat every 8 tasks)
    got = [[None] * ROUNDS for _ in range(W)]
    errors = []
    streams = [zk.Stream() for _ in range(W)]
    barrier = threading.Barrier(W)

    def work(w):
        try:
            su, cms = workers[w]
            barrier.wait()
            for rep in range(2):                                          # twice: the second pass reuses pooled blocks freed by every thread
                for r, cm in enumerate(cms):
                    got[w][r] = su.gen(cm, stream=streams[w].handle)
                    assert got[w][r] == alone[w][r], "worker %d round %d pass %d" % (w, r, rep)
        except BaseException as e:                                        # noqa: BLE001 -- reported by the main thread
            errors.append((w, repr(e)))

    threads = [threading.Thread(target=work, args=(w,)) for w in range(W)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a prover thread hangs"
    assert not errors, errors
    zk.lib().zk_dev_sync()
    assert got == alone
    for st in streams: st.free()
