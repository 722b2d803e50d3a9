"""Test backends for the C ABI: 'hip' = the real libccd_hip.so on cuda:0 (GPU box), 'sim' = the same kernel
sources compiled against tests/hipsim (CPU SIMT executor) so kernel logic can be checked without a GPU."""
import ctypes
import os
import subprocess

import torch

from ccd_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_SO = os.path.join(HERE, "hipsim", "libccd_sim.so")


def _sim_stale():
    if not os.path.isfile(SIM_SO):
        return True
    t = os.path.getmtime(SIM_SO)
    roots = [os.path.join(HERE, "hipsim"), os.path.join(HERE, "..", "ccd_amd", "csrc"), os.path.join(HERE, "..", "include")]
    for r in roots:
        for d, _, fs in os.walk(r):
            for f in fs:
                if f.endswith((".h", ".cpp", ".hip")) and os.path.getmtime(os.path.join(d, f)) > t:
                    return True
    return False


class Backend:
    def __init__(self, kind):
        self.kind = kind
        self.device = torch.device("cuda:0" if kind == "hip" else "cpu")

    def __enter__(self):
        self._saved = (_lib._handle, _lib._stream_override)
        if self.kind == "sim":
            if _sim_stale():
                subprocess.check_call([os.path.join(HERE, "hipsim", "build.sh")])
            _lib._handle = _lib.bind(ctypes.CDLL(SIM_SO))
            _lib._stream_override = 0
        else:
            _lib._handle, _lib._stream_override = None, None
            _lib.get()
        return self

    def __exit__(self, *a):
        _lib._handle, _lib._stream_override = self._saved

    def sync(self):
        if self.kind == "hip":
            torch.cuda.synchronize()
