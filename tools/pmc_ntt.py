"""Workload for tools/gpu_round.sh profiles / sq: calibration copy (1 GiB read + 1 GiB write) then 2^24 NTT fwd+inv x4,
and one 20-column 2^20 -> 2^21 LDE + Merkle tree (the prover's stage-1 shape)."""
import sys, pathlib
import numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
a = torch.arange(1 << 27, dtype=torch.int64, device="cuda")
for _ in range(3):
    b = a.clone()
torch.cuda.synchronize()
del a, b
n = 1 << 24
x = zk.DevArray.from_host(np.arange(n, dtype=np.uint64))
y, t = zk.DevArray(n), zk.DevArray(n)
L = zk.lib()
for _ in range(4):
    zk._check(L.zk_gl_ntt_dev(x.ptr, y.ptr, t.ptr, 1, 24, 0, None))
    zk._check(L.zk_gl_ntt_dev(y.ptr, x.ptr, t.ptr, 1, 24, 1, None))
L.zk_dev_sync()
w, nb = 20, 20
src = zk.DevArray.from_host(np.arange((1 << nb) * w, dtype=np.uint64))
dst, tmp = zk.DevArray((2 << nb) * w), zk.DevArray((2 << nb) * w)
zk._check(L.zk_gl_lde_dev(src.ptr, w, nb, dst.ptr, tmp.ptr, nb + 1, None))
tree = zk.MerkleTreeGL(); tree.merkelize_dev(dst.ptr, w, 2 << nb)
L.zk_dev_sync()
print("done")
