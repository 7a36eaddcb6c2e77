import sys, json, pathlib, importlib, functools
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import zkgpu_loader, synth_pil
zk = zkgpu_loader.load(); zk.init(0)
stark = importlib.import_module("eigen_zkvm_amd.stark")
def wrap(mod, name):
    f = getattr(mod, name)
    @functools.wraps(f)
    def g(*a, **k):
        r = f(*a, **k)
        rc = zk.lib().zk_dev_sync()
        print("ok", name, rc, flush=True)
        return r
    setattr(mod, name, g)
for n in ("lev", "xdivxsub", "fri_fold", "fri_transpose", "x_table", "zh_inv", "qsplit"):
    wrap(stark, n)
for n in ("merkelize_dev", "get_group_proof", "root"):
    wrap(zk.MerkleTreeGL, n)
for n in ("put_dev", "get_field_dev", "get_permutations"):
    wrap(zk.TranscriptGL, n)
wrap(zk.Program, "run")
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 10
d = json.load(open(ROOT / "tests" / "golden" / "widefib_w10.program.json"))
d["starkinfo"]["exp2pol"] = {int(k): v for k, v in d["starkinfo"]["exp2pol"].items()}
info = synth_pil.rescale(d["starkinfo"], nbits); ss = synth_pil.stark_struct(nbits)
cm = synth_pil.wide_fib_trace(nbits, 10); const = synth_pil.const_trace(nbits)
setup = stark.StarkSetup(const, info, d["program"], ss); print("setup ok", flush=True)
for r in range(3):
    proof = stark.stark_gen(cm, setup); print("proof", r, proof["root1"][:1], flush=True)
