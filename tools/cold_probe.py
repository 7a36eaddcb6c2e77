"""Where a fresh process spends its first 2^nbits-row PoseidonG proof (round 6, one-shot path): python tools/cold_probe.py [nbits]
Prints the setup split, then per proof the wall time and the stage timers -- the first proof against the steady state -- and what hipMalloc
itself costs for the sizes a proof asks for (through zk_dev_alloc: first allocation vs a reuse from the pool)."""
import json, os, pathlib, sys, time
os.environ.setdefault("ZK_STARK_TIMING", "quiet")
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import importlib
import eigen_zkvm_amd as zk
t_imp = time.perf_counter()
zk.init(0)
print("zk.init: %.3f s" % (time.perf_counter() - t_imp), flush=True)
import poseidong as PG
stark = importlib.import_module("eigen_zkvm_amd.stark")
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lib = zk.lib()
for gb in (1, 8):
    for rep in range(2):
        t0 = time.perf_counter(); a = zk.DevArray(gb << 27); lib.zk_dev_sync(); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); a.free() if hasattr(a, "free") else None; del a; lib.zk_dev_sync(); t2 = time.perf_counter() - t0
        print("zk_dev_alloc %d GiB (%s): alloc %.1f ms, free %.1f ms" % (gb, "first" if rep == 0 else "again", t1 * 1e3, t2 * 1e3), flush=True)
lib.zk_dev_trim()
ss = PG.stark_struct(nbits); pj = json.dumps(PG.program(nbits))
const, cm = PG.consts(nbits), PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits)
t0 = time.perf_counter()
setup = stark.NativeStarkSetup(const, pj, json.dumps(ss)); lib.zk_dev_sync()
print("setup: %.3f s %s" % (time.perf_counter() - t0, json.dumps(setup.setup_timing())), flush=True)
t0 = time.perf_counter(); d_cm = zk.DevArray.from_host(cm); lib.zk_dev_sync()
print("trace upload (%.2f GB): %.1f ms" % (cm.nbytes / 1e9, (time.perf_counter() - t0) * 1e3), flush=True)
for k in range(3):
    t0 = time.perf_counter(); txt = setup.gen_json(d_cm); dt = time.perf_counter() - t0
    print("proof %d: %.1f ms  stages %s" % (k, dt * 1e3, json.dumps(setup.last_timing())), flush=True)
