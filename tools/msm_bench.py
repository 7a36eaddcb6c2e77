"""BN254 G1 MSM timing (device-resident inputs): python tools/msm_bench.py [logn ...]"""
import sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import eigen_zkvm_amd

zk = eigen_zkvm_amd; zk.init(0)
curve = "bn254"
args = sys.argv[1:]
if args and args[0] in ("bn254", "bls12_381"):
    curve = args.pop(0)
group = "g1"
if args and args[0] in ("g1", "g2"):
    group = args.pop(0)
print("curve", curve, group)
for logn in [int(a) for a in args] or [16, 20, 22]:
    n = 1 << logn
    rng = np.random.default_rng(logn)
    k = rng.integers(1, 2**64, size=n, dtype=np.uint64)
    scal = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); scal[:, 3] &= np.uint64((1 << 60) - 1)
    t = time.perf_counter(); db = zk.g1_mul_generator(zk.DevArray.from_host(k), curve, group=group); zk.lib().zk_dev_sync()
    print(f"  bases on device: {time.perf_counter()-t:.3f} s")
    ds = zk.DevArray.from_host(scal.reshape(-1))
    zk.msm_g1_dev(db, ds, n, curve, group=group)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); o = zk.msm_g1_dev(db, ds, n, curve, group=group); zk.lib().zk_dev_sync(); ts.append(time.perf_counter() - t)
    print(f"msm 2^{logn}: {min(ts)*1e3:.2f} ms  {n/min(ts)/1e6:.2f} Mpts/s", flush=True)
