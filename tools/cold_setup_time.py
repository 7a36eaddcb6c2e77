"""Cold StarkSetup::new of the PoseidonG PIL (no code-object cache): python tools/cold_setup_time.py [nbits]   -> zk_stark_setup_timing"""
import json, os, pathlib, sys, time
os.environ["ZK_JIT_CACHE"] = "off"
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import eigen_zkvm_amd as zk, importlib
zk.init(0)
import poseidong as PG
stark = importlib.import_module("eigen_zkvm_amd.stark")
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ss = PG.stark_struct(nbits)
pj = json.dumps(PG.program(nbits))
const = PG.consts(nbits)
t0 = time.perf_counter()
s = stark.NativeStarkSetup(const, pj, json.dumps(ss))
zk.lib().zk_dev_sync()
print("cold setup 2^%d: %.2f s" % (nbits, time.perf_counter() - t0), json.dumps(s.setup_timing()), flush=True)
