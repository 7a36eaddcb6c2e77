"""Host-side mirror of compressor12 exec (recursion/src/compressor12/compressor12_exec.rs:17-103) over libzkgpu's
C ABI: the PlonkAdd sums and the s_map gather run on the device and leave the committed trace in HBM."""
import numpy as np

from . import DevArray, ZkError, _check, lib


class Compressor12Exec:
    def __init__(self, exec_text, n_witness):
        b = exec_text.encode() if isinstance(exec_text, str) else bytes(exec_text)
        self.n_witness = n_witness
        self._h = lib().zk_c12_exec_new(b, len(b), n_witness)
        if not self._h:
            raise ZkError(lib().zk_last_error().decode())
        self.depth = lib().zk_c12_exec_depth(self._h)

    def run(self, witness, n_rows, stream=0):
        """witness: u64 host array or DevArray of n_witness words -> DevArray [n_rows][12] (row-major), the .cm content"""
        d_w = witness if isinstance(witness, DevArray) else DevArray.from_host(np.ascontiguousarray(witness, dtype=np.uint64))
        cm = DevArray(max(n_rows * 12, 1))
        _check(lib().zk_c12_exec_dev(self._h, d_w.ptr, d_w.n, n_rows, cm.ptr, stream))
        return cm

    def free(self):
        if self._h:
            lib().zk_c12_exec_free(self._h); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
