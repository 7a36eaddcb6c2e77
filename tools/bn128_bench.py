"""MerkleTreeBN128 timing: python tools/bn128_bench.py [log_height width]..."""
import sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0); zk.bn128_init()
args = [int(a) for a in sys.argv[1:]] or [16, 12, 18, 12, 20, 12, 18, 48]
for lh, w in zip(args[::2], args[1::2]):
    h = 1 << lh
    rng = np.random.default_rng(1)
    d = zk.DevArray.from_host(rng.integers(0, 0xFFFFFFFF00000001, size=h * w, dtype=np.uint64))
    ts = []
    for _ in range(3):
        t = time.perf_counter(); tr = zk.MerkleTreeBN128(); tr.merkelize_dev(d.ptr, w, h); zk.lib().zk_dev_sync(); ts.append(time.perf_counter() - t); root = tr.root(); tr.free()
    nb = (w - 1) // 3 + 1 if w > 4 else 0
    leaf_perms = (nb + 15) // 16
    n, nodes = h, 0
    while n > 1:
        n = (n - 1) // 16 + 1; nodes += n
    print(f"merkelize_bn128 2^{lh} x {w}: {min(ts)*1e3:.2f} ms  ({leaf_perms} leaf perms/row of t<={min(17, nb + 1)}, {nodes} t=17 node perms)  root0={root[0]}", flush=True)
