"""Raw timing of the 2^nbits NTT (no correctness check): python tools/ntt_time.py [nbits] [n_pols]"""
import sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 24
npols = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = (1 << nbits) * npols
x = zk.DevArray.from_host(np.arange(n, dtype=np.uint64)); y, t = zk.DevArray(n), zk.DevArray(n)
L = zk.lib()
for inv in (0, 1):
    for _ in range(3): L.zk_gl_ntt_dev(x.ptr, y.ptr, t.ptr, npols, nbits, inv, None)
    L.zk_dev_sync(); t0 = time.perf_counter()
    for _ in range(20): L.zk_gl_ntt_dev(x.ptr, y.ptr, t.ptr, npols, nbits, inv, None)
    L.zk_dev_sync(); dt = (time.perf_counter() - t0) / 20
    p = L.zk_gl_ntt_passes(nbits)
    print(f"nbits={nbits} np={npols} inv={inv}: {dt*1e6:.1f} us/transform, {dt*1e6/p:.1f} us/pass, {n/dt/1e9:.1f} GElem/s, pass traffic {16*n*p/dt/1e12:.2f} TB/s", flush=True)
