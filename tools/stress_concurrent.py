"""Eight provers (Fibonacci 2^10, compressor-shaped 2^15 and 2^18) on eight host threads and streams, 192 proofs: each must equal the
proof its setup gives alone.  python tools/stress_concurrent.py"""
import json, pathlib, sys, threading, time
ROOT = pathlib.Path(__file__).resolve().parent.parent; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
import importlib, aggregation_workload as AW
stark = importlib.import_module("eigen_zkvm_amd.stark")
W, ROUNDS, REPS = 8, 4, 6
circ = {k: AW.Circuit(AW.STRUCTS[k]["nBits"]) for k in ("c12", "r1")}
consts = {"fib": AW.fib_consts(), "c12": circ["c12"].consts, "r1": circ["r1"].consts}
kinds = ["fib", "c12", "r1"]
workers = []
for w in range(W):
    kind = kinds[w % 3]
    su = stark.NativeStarkSetup(consts[kind], json.dumps(AW.program(kind)), json.dumps(AW.STRUCTS[kind]))
    cms = [zk.DevArray.from_host(AW.fib_trace(10 * w + r) if kind == "fib" else circ[kind].witness(10 * w + r)) for r in range(ROUNDS)]
    workers.append((su, cms))
alone = [[su.gen_json(cm) for cm in cms] for su, cms in workers]
errors = []; streams = [zk.Stream() for _ in range(W)]
def work(w):
    try:
        su, cms = workers[w]
        for rep in range(REPS):
            for r, cm in enumerate(cms):
                z = su.gen_json(cm, streams[w].handle)
                if z != alone[w][r]: errors.append((w, r, rep)); return
    except BaseException as e: errors.append((w, repr(e)))
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(w,)) for w in range(W)]
for t in th: t.start()
for t in th: t.join()
print("stress: %d proofs in %.2f s, errors: %s" % (W * ROUNDS * REPS, time.perf_counter() - t0, errors))
