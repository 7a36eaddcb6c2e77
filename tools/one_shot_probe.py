"""The one-shot path outside bench.py: python tools/one_shot_probe.py [nbits] [runs] -- writes the PoseidonG inputs as files, then runs
`zkgpu_prove.py stark_prove` as a fresh process `runs` times (the first compiles the step programs, the later ones find the code objects
on disk) and prints each child's wall time and split."""
import json, os, pathlib, shutil, subprocess, sys, tempfile, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import poseidong as PG
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 22
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = pathlib.Path(tempfile.mkdtemp(prefix="zk_one_shot_"))
PG.consts(nbits).tofile(d / "c.const"); PG.trace(nbits, None, PG.FIRST_ZERO, seed=nbits).tofile(d / "c.cm")
(d / "pil.json").write_text(json.dumps(PG.pil(nbits))); (d / "ss.json").write_text(json.dumps(PG.stark_struct(nbits)))
cmd = [sys.executable, str(ROOT / "tools" / "zkgpu_prove.py"), "stark_prove", "-s", str(d / "ss.json"), "-p", str(d / "pil.json"),
       "--o", str(d / "c.const"), "--m", str(d / "c.cm"), "--i", str(d / "zkin.json")]
for k in range(runs):
    t0 = time.perf_counter(); r = subprocess.run(cmd + sys.argv[3:], capture_output=True, text=True); dt = time.perf_counter() - t0
    print("run %d: %.3f s rc=%d" % (k, dt, r.returncode)); print("   ", "\n    ".join(l for l in r.stderr.splitlines() if "timing" in l or "cache" in l or "rror" in l))
cache = os.path.join(os.environ.get("HOME", "/nonexistent"), ".cache", "zkgpu")
print("cache dir", cache, sorted(os.listdir(cache))[:5] if os.path.isdir(cache) else "absent")
shutil.rmtree(d, ignore_errors=True)
