"""Scalar-field Merkle tree timing on the device: python tools/fr_merkle_time.py <field> <log2 height> <width> [reps]
prints ms per tree (HIP events would be finer; trees here take milliseconds) and the root, so that two builds or two settings of a
tuning knob (ZK_FRHASH_REG_MAX, ZK_FR_LEVEL_COOP) can be compared for speed AND for equality."""
import pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import eigen_zkvm_amd as zk
zk.init(0)
field, lg, width = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
zk.bn128_init(field=field)
h = 1 << lg
rng = np.random.default_rng(7)
rows = rng.integers(0, 0xFFFFFFFF00000001, size=h * width, dtype=np.uint64)
d = zk.DevArray.from_host(rows)
t = zk.MerkleTreeBN128(field)
t.merkelize_dev(d.ptr, width, h); root = t.root(); t.free()
zk.synchronize() if hasattr(zk, "synchronize") else None
t0 = time.perf_counter()
for _ in range(reps):
    t = zk.MerkleTreeBN128(field); t.merkelize_dev(d.ptr, width, h); r2 = t.root(); t.free()
ms = (time.perf_counter() - t0) / reps * 1e3
print(f"{field} 2^{lg} x {width}: {ms:.3f} ms per tree, root {[hex(int(v)) for v in root]}", flush=True)
assert (r2 == root).all()
