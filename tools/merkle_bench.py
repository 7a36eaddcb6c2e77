"""Merkle-GL timing: python tools/merkle_bench.py [log_height width]..."""
import sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
args = [int(a) for a in sys.argv[1:]] or [21, 20, 22, 20, 21, 12, 21, 6]
for lh, w in zip(args[::2], args[1::2]):
    h = 1 << lh
    rng = np.random.default_rng(1)
    d = zk.DevArray.from_host(rng.integers(0, 0xFFFFFFFF00000001, size=h * w, dtype=np.uint64))
    ts = []
    for _ in range(4):
        t = time.perf_counter(); tr = zk.MerkleTreeGL(); tr.merkelize_dev(d.ptr, w, h); zk.lib().zk_dev_sync(); ts.append(time.perf_counter() - t); root = tr.root(); tr.free()
    bs = max(8, (w + 3) // 4); nb = (w + bs - 1) // bs
    perms = sum((min(bs, w - b * bs) + 7) // 8 for b in range(nb) if min(bs, w - b * bs) > 4) + ((nb * 4 + 7) // 8 if nb > 1 else 0) if w > 4 else 0
    total = (perms + 1) * h
    print(f"merkelize 2^{lh} x {w}: {min(ts)*1e3:.2f} ms  {total/min(ts)/1e9:.3f} Gperm/s  root0={root[0]}", flush=True)
