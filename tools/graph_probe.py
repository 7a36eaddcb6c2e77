"""Does replaying a 2^24 forward+inverse NTT as a captured HIP graph close the gaps between its six dependent launches?"""
import ctypes as C, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
torch.cuda.set_device(0)
import eigen_zkvm_amd, oracle_lib
zk = eigen_zkvm_amd; zk.init(0)
lib = zk.lib(); vp = C.c_void_p
nbits = 24; n = 1 << nbits
x = torch.from_numpy(oracle_lib.splitmix64_stream(1, n).view(np.int64)).cuda()
X = torch.empty_like(x); y = torch.empty_like(x); tmp = torch.empty_like(x)
def step(stream):
    rc = lib.zk_gl_ntt_dev(vp(x.data_ptr()), vp(X.data_ptr()), vp(tmp.data_ptr()), 1, nbits, 0, vp(stream))
    rc |= lib.zk_gl_ntt_dev(vp(X.data_ptr()), vp(y.data_ptr()), vp(tmp.data_ptr()), 1, nbits, 1, vp(stream))
    assert rc == 0, lib.zk_last_error()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): step(s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step(s.cuda_stream)
    torch.cuda.synchronize(); print("plain launches: %.1f us/step" % ((time.perf_counter() - t0) / 50 * 1e6), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step(s.cuda_stream)
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); print("graph replay:   %.1f us/step" % ((time.perf_counter() - t0) / 50 * 1e6), flush=True)
    assert torch.equal(x, y)
    g10 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g10, stream=s):
        for _ in range(10): step(s.cuda_stream)
    torch.cuda.synchronize(); g10.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g10.replay()
    torch.cuda.synchronize(); print("graph of 10:    %.1f us/step" % ((time.perf_counter() - t0) / 50 * 1e6), flush=True)
