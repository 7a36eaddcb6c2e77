"""CPU-side checks of the boundary: libzkgpu.so loads without a GPU and exports every symbol
include/zkgpu.h declares (no compute calls here)."""
import ctypes, pathlib, re

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "zkgpu.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(zk):
    lib = ctypes.CDLL(str(zk.LIB_PATH))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/zkgpu.h but not exported"
    assert sorted(zk.EXPORTS) == names


def test_constants_without_gpu(zk):
    lib = zk.lib()
    assert lib.zk_gl_modulus() == 0xFFFFFFFF00000001
    assert lib.zk_gl_root_of_unity(32) == pow(7, 2**32 - 1, 0xFFFFFFFF00000001)
    assert lib.zk_gl_root_of_unity(1) == 0xFFFFFFFF00000000
    assert lib.zk_merkle_n_nodes(256) == 511 and lib.zk_merkle_n_nodes(33) == 34 + 18 + 10 + 6 + 4 + 2 + 1
    assert lib.zk_gl_ntt_passes(24) == 3 and lib.zk_gl_ntt_passes(8) == 1 and lib.zk_gl_ntt_passes(3) == 1


def test_no_cpu_fallback_in_product():
    """The product package must not reference the oracle."""
    for f in (ROOT / "eigen-zkvm_amd").rglob("*"):
        if f.is_file() and f.suffix in (".py", ".hip", ".h", ".cuh", ".cpp"):
            assert "oracle" not in f.read_text().replace("no CPU fallback", ""), f
