"""One small circuit proved in a loop (for rocprofv3 --kernel-trace --hip-trace --stats): python tools/small_proof_probe.py fib|c12|r1 [n]"""
import json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
import aggregation_workload as AW
kind = sys.argv[1] if len(sys.argv) > 1 else "fib"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
P = AW.pool(zk, workers=1)
i = {"fib": 0, "c12": 1, "r1": 2}[kind]
inp = P.task_inputs(0)[i]
for _ in range(3): P.prove([inp], 0)
P.sync(); t0 = time.perf_counter()
for _ in range(n): P.prove([inp], 0)
P.sync(); print(f"{kind}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per proof", flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "timing":                         # the library's own stage report of one more proof
    import os
    os.environ["ZK_STARK_TIMING"] = "1"
    P.prove([inp], 0); P.sync()
    print(json.dumps(P.sets[0][kind].last_timing()), flush=True)
