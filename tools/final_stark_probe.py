"""The final STARK of BASELINE config 5 (final.starkStruct.bls12381.json: 2^16 rows, BLS12381 hashing) proved in a loop, for
rocprofv3 --kernel-trace --hip-trace --stats: python tools/final_stark_probe.py [n]"""
import json, pathlib, sys, time, importlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
import aggregation_workload as AW, poseidong as PG
stark = importlib.import_module("eigen_zkvm_amd.stark")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ss = {"nBits": 16, "nBitsExt": 17, "nQueries": 8, "verificationHashType": "BLS12381", "steps": [{"nBits": 17}, {"nBits": 7}, {"nBits": 3}]}
circ = AW.Circuit(16)
setup = stark.NativeStarkSetup(circ.consts, json.dumps(PG.native_program(AW.c12_pil(16), ss)), json.dumps(ss))
d_cm = zk.DevArray.from_host(circ.witness(primary=[1, 2, 3, 4] + [0] * 12))
setup.gen_json(d_cm)
t0 = time.perf_counter()
for _ in range(n): setup.gen_json(d_cm)
print(f"final STARK: {(time.perf_counter() - t0) / n * 1e3:.1f} ms per proof", flush=True)
