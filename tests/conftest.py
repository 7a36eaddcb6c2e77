"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol checks (CPU only).
`-m gpu`       : parity tests proper, HIP path (through the C ABI) vs the oracle.
"""
import os, sys, pathlib
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(ROOT / "tests" / "golden" / "kat.json") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def zk():
    """The product library through its C ABI (ctypes binding in eigen-zkvm_amd/__init__.py)."""
    import zkgpu_loader
    return zkgpu_loader.load()
