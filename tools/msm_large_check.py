"""Large multi-scalar sums against the closed form [sum s_i k_i mod r]G (bases P_i = [k_i]G made on the device) + timing:
python tools/msm_large_check.py [curve] logn ...   (needs a GPU; the host-side sum of 2^24 256-bit products takes ~20 s)"""
import pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import eigen_zkvm_amd as zk, oracle_lib
zk.init(0)
orc = oracle_lib.load()
args = sys.argv[1:]
curve = args.pop(0) if args and args[0] in ("bn254", "bls12_381") else "bn254"
cv = orc.curve(curve); R = cv.r; nl = {"bn254": 4, "bls12_381": 6}[curve]
w = lambda x: np.array([(x >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64)
for logn in [int(a) for a in args]:
    n = (1 << logn) + (3 if logn % 2 else 0)                                 # odd sizes too
    rng = np.random.default_rng(logn)
    k = rng.integers(1, 2**64, size=n, dtype=np.uint64)
    scal = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); scal[:, 3] &= np.uint64((1 << 60) - 1)
    db = zk.g1_mul_generator(zk.DevArray.from_host(k), curve); ds = zk.DevArray.from_host(scal.reshape(-1))
    out = zk.msm_g1_dev(db, ds, n, curve); zk.lib().zk_dev_sync()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); out = zk.msm_g1_dev(db, ds, n, curve); zk.lib().zk_dev_sync(); ts.append(time.perf_counter() - t)
    t0 = time.perf_counter()
    acc = 0
    for lo in range(0, n, 1 << 20):                                          # sum s_i k_i mod r in slices (object arithmetic)
        s4 = scal[lo:lo + (1 << 20)].astype(object)
        sv = s4[:, 0] + (s4[:, 1] << 64) + (s4[:, 2] << 128) + (s4[:, 3] << 192)
        acc = (acc + int((sv * k[lo:lo + (1 << 20)].astype(object)).sum())) % R
    exp, _ = cv.scalar_mul(cv.generator(), w(acc))
    ok = np.array_equal(out.to_host()[:2 * nl], exp)
    print("msm %s n=%d: %.2f ms  %.1f Mpts/s  closed form %s (host check %.0f s)" % (curve, n, min(ts) * 1e3, n / min(ts) / 1e6, "OK" if ok else "MISMATCH", time.perf_counter() - t0), flush=True)
