import sys, pathlib, importlib, numpy as np
ROOT = pathlib.Path("/root/repo"); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import zkgpu_loader
zk = zkgpu_loader.load(); zk.init(0)
dev = importlib.import_module("eigen_zkvm_amd.groth16")
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x = zk.DevArray.from_host(np.arange(4 << logn, dtype=np.uint64) % 1000)
for _ in range(3): dev.fr_ntt(x, "BN128")
zk.lib().zk_dev_sync()
