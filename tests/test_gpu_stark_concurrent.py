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


_COLD = r'''
import json, sys, threading, importlib, pathlib
ROOT = pathlib.Path(sys.argv[1]); mode = sys.argv[2]
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import zkgpu_loader
zk = zkgpu_loader.load(); zk.init(0)
import aggregation_workload as AW
stark = importlib.import_module("eigen_zkvm_amd.stark")
W = 4
circ = AW.Circuit(10)
ss = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}
import poseidong
prog = json.dumps(poseidong.native_program(AW.c12_pil(10), ss))
sets = [stark.NativeStarkSetup(circ.consts, prog, json.dumps(ss)) for _ in range(W)]      # c12 shape: tree 2 is the empty-section tree
cms = [zk.DevArray.from_host(circ.witness(w)) for w in range(W)]
if mode == "msm":      # asynchronous null-stream work whose scratch goes back to the pool while still in flight, then streamed proofs
    import numpy as np
    n = 1 << 14
    k = np.arange(1, n + 1, dtype=np.uint64)
    sc = np.zeros((n, 4), np.uint64); sc[:, 0] = k
    d_k = zk.DevArray.from_host(k); d_s = zk.DevArray.from_host(sc.reshape(-1)); d_b = zk.DevArray(n * 8); d_out = zk.DevArray(9)
    assert zk.lib().zk_g1_bn254_mul_generator_dev(d_k.ptr, n, d_b.ptr, None) == 0
    for _ in range(3): assert zk.lib().zk_msm_g1_bn254_dev(d_b.ptr, d_s.ptr, n, d_out.ptr, None) == 0
streams = [zk.Stream() for _ in range(W)]
got, errs = [None] * W, []
bar = threading.Barrier(W)
def work(w):
    try:
        bar.wait(); got[w] = sets[w].gen_json(cms[w], streams[w].handle)      # the very first proofs of the process, all at once
    except BaseException as e: errs.append(repr(e))
ts = [threading.Thread(target=work, args=(w,)) for w in range(W)]
[t.start() for t in ts]; [t.join(600) for t in ts]
assert not errs, errs
zk.lib().zk_dev_sync()
alone = [sets[w].gen_json(cms[w]) for w in range(W)]                           # now warm, one at a time, default stream
assert got == alone, [i for i in range(W) if got[i] != alone[i]]
print("cold-start ok", mode)
'''


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "msm"])
def test_cold_start_concurrent_proofs_in_a_fresh_process(mode):
    """Round-2 advisor findings: (1) the all-zero-subtree digests were built lazily by the first prover to need them and
    published before they were in memory -- a second prover on another stream could read them uninitialised, but only on a
    cold start, which the test above hides by proving sequentially first; (2) pool blocks freed while the null stream was the
    only stream carried no event.  A fresh process whose first proofs run concurrently (after asynchronous null-stream sums
    in the "msm" variant) must give the proofs the same setups give alone."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", _COLD, str(ROOT), mode], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "cold-start ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
