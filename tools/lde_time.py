"""LDE timing: python tools/lde_time.py nbits np [np ...]   (the plan is inverse + one 2N-point forward transform over a half-zero input)"""
import sys, time, pathlib, ctypes
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
nbits = int(sys.argv[1])
for np_ in [int(a) for a in sys.argv[2:]]:
    n = 1 << nbits
    src = zk.DevArray(n * np_); dst = zk.DevArray(2 * n * np_); tmp = zk.DevArray(2 * n * np_)
    assert zk.lib().zk_dev_fill_splitmix(src.ptr, n * np_, 7, None) == 0
    ts = []
    for _ in range(5):
        zk.lib().zk_dev_sync(); t0 = time.perf_counter()
        assert zk.lib().zk_gl_lde_dev(src.ptr, np_, nbits, dst.ptr, tmp.ptr, nbits + 1, None) == 0
        zk.lib().zk_dev_sync(); ts.append(time.perf_counter() - t0)
    print(f"lde 2^{nbits} -> 2^{nbits + 1} x {np_}: {min(ts) * 1e3:.2f} ms", flush=True)
    src.free(); dst.free(); tmp.free()
