import sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import zkgpu_loader, oracle_lib
zk = zkgpu_loader.load(); zk.init(0)
orc = oracle_lib.load()
for curve in ("bn254", "bls12_381"):
    cv = orc.curve(curve, g2=True)
    t = time.time()
    k = np.array([1, 2, 3, 5], np.uint64)
    b = zk.g1_mul_generator(zk.DevArray.from_host(k), curve, group="g2").to_host().reshape(4, -1)
    print(curve, "generator kernel", time.time() - t, flush=True)
    for i in range(4):
        exp, _ = cv.scalar_mul(cv.generator(), np.array([int(k[i]), 0, 0, 0], np.uint64))
        print(i, np.array_equal(b[i], exp), flush=True)
    t = time.time()
    got, inf = zk.msm_g1(b[:2].reshape(-1), np.array([7, 0, 0, 0, 9, 0, 0, 0], np.uint64), curve, group="g2")
    exp, _ = cv.scalar_mul(cv.generator(), np.array([7 + 18, 0, 0, 0], np.uint64))
    print("msm n=2", time.time() - t, inf, np.array_equal(got, exp), flush=True)
