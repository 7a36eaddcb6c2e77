"""`import eigen_zkvm_amd` from the repo root: the package directory is named `eigen-zkvm_amd` (the hyphen is the task's
spelling), which Python cannot import by name -- this module becomes that package (same namespace, `__path__` on the
directory, so `eigen_zkvm_amd.stark`, `.groth16`, `.compressor12`, `.aggregation` import as its submodules)."""
import pathlib as _pathlib

_pkg = _pathlib.Path(__file__).resolve().parent / "eigen-zkvm_amd"
__path__ = [str(_pkg)]
__file__ = str(_pkg / "__init__.py")
exec(compile((_pkg / "__init__.py").read_text(), __file__, "exec"))
