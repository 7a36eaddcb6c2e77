"""Cost probe for batched-affine bucket additions (round-2 verdict, item 6): a variant build of libzkgpu with -DZK_MSM_UBENCH
(tools/build_variant.sh msmub msm.hip "-DZK_MSM_UBENCH") times the shipped XYZZ mixed addition against affine additions that share
one inversion per lane (csrc/msm_impl.hip.h ubench_*): python tools/msm_affine_ubench.py"""
import ctypes, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
os.environ.setdefault("ZKGPU_LIB", str(ROOT / "eigen-zkvm_amd" / "variants" / "libzkgpu_msmub.so"))
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import eigen_zkvm_amd
zk = eigen_zkvm_amd; zk.init(0)
lib = ctypes.CDLL(os.environ["ZKGPU_LIB"])
lib.zk_msm_ubench_affine.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
lanes = 1 << 17                                                          # two waves per SIMD, as the accumulation kernel runs
for K in (32, 64, 128, 256):
    total = lanes * K                                                    # points: distinct multiples of the generator
    k = np.arange(1, total + 1, dtype=np.uint64)
    d_k = zk.DevArray.from_host(k); d_b = zk.DevArray(total * 8)
    assert zk.lib().zk_g1_bn254_mul_generator_dev(d_k.ptr, total, d_b.ptr, None) == 0
    zk.lib().zk_dev_sync()
    for mode in (0, 1):
        assert lib.zk_msm_ubench_affine(d_b.ptr, lanes, K, mode) == 0
    d_k.free(); d_b.free(); zk.lib().zk_dev_trim()
