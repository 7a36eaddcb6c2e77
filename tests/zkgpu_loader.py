"""Loads the product's ctypes binding.  The package directory is `eigen-zkvm_amd` (hyphen, as
the task names it), so it is imported by path under the module name `eigen_zkvm_amd`."""
import importlib.util, pathlib, sys

ROOT = pathlib.Path(__file__).resolve().parent.parent


def load():
    if "eigen_zkvm_amd" in sys.modules:
        return sys.modules["eigen_zkvm_amd"]
    pkg = ROOT / "eigen-zkvm_amd"
    spec = importlib.util.spec_from_file_location("eigen_zkvm_amd", pkg / "__init__.py",
                                                  submodule_search_locations=[str(pkg)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["eigen_zkvm_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
