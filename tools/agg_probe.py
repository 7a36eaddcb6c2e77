"""Where the time of BASELINE config 5's small proofs goes: each circuit alone, then k of the same at once.
python tools/agg_probe.py [workers]"""
import json, pathlib, sys, time, threading
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
import aggregation_workload as AW
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = AW.pool(zk, workers=W)
ins = [P.task_inputs(t) for t in range(W)]
for kind_i, kind in enumerate(("fib", "c12", "r1")):
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(4): P.prove([ins[0][kind_i]], 0)
        P.sync(); dt1 = (time.perf_counter() - t0) / 4
    t0 = time.perf_counter()
    for _ in range(4): P.sets[0][kind].gen_json(ins[0][kind_i][1], P.streams[0].handle)
    P.sync(); dtj = (time.perf_counter() - t0) / 4
    z = P.sets[0][kind].gen_json(ins[0][kind_i][1])
    for rep in range(2):
        t0 = time.perf_counter()
        P._spread([[i[kind_i]] for i in ins] * 2, P.prove)
        P.sync(); dtw = (time.perf_counter() - t0) / (2 * W)
    print(f"{kind}: alone {dt1*1e3:.2f} ms/proof (gen_json only {dtj*1e3:.2f}), {W} at once {dtw*1e3:.2f} ms/proof, zkin {len(z)/1e3:.0f} kB", flush=True)
