"""Latency of one cooperative Poseidon permutation (16 lanes): a transcript absorbing n words is a chain of n / 8 dependent permutations.
python tools/coop_perm_time.py [n_perms]"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = zk.DevArray.from_host(np.arange(8 * n, dtype=np.uint64))
best = 1e9
for _ in range(4):
    t = zk.TranscriptGL()
    zk.lib().zk_dev_sync(); t0 = time.perf_counter()
    t.put_dev(d)
    zk.lib().zk_dev_sync(); best = min(best, time.perf_counter() - t0)
print(f"coop_perm: {best / n * 1e6:.2f} us per permutation ({n} chained, tr_put_kernel)", flush=True)
