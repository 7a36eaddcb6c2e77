"""Kept for the tests' `zk` fixture: the product's ctypes binding is `import eigen_zkvm_amd` with the repo root on sys.path
(eigen_zkvm_amd.py there maps the name onto the `eigen-zkvm_amd` directory)."""
import pathlib, sys

ROOT = pathlib.Path(__file__).resolve().parent.parent


def load():
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    import eigen_zkvm_amd
    return eigen_zkvm_amd
