"""CU-masked HIP streams: student and teacher forward passes side by side on disjoint sets of compute units.

The two backbone forward passes of an iteration (train.py:232-233) are independent until the loss.  Run one after the other,
every CU of the chip is in the same phase of the same kernel at the same time (an MFMA phase, then an HBM phase: docs/LAB_NOTEBOOK.md
section 4d); run on two streams whose queues are restricted to disjoint CU sets (hipExtStreamCreateWithCUMask), the two
partitions execute different kernels, so the phases of one fall into the other's gaps.

Mask layout on a multi-XCD part (KFD's symmetric mapping): bit i of the mask is compute unit i // 8 of XCD i % 8.  Both
partitions keep CUs on EVERY XCD (a queue with no CU on some XCD is not something the dispatcher's round-robin over XCDs is
documented to survive), so the kernels' `blockIdx % 8 = XCD` placement stays valid inside a partition.

`partition(n_teacher)` -> (student stream, teacher stream, CUs of the student, CUs of the teacher) or None off the GPU.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HIP = None
_CACHE = {}


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
        _HIP.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32,
                                                      ctypes.POINTER(ctypes.c_uint32)]
        _HIP.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    return _HIP


def masked_stream(bits, device):
    """A stream of `device` whose kernels may only run on the compute units whose bit is set in `bits` (iterable of bit indices)."""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * len(words))(*words)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = _hip().hipExtStreamCreateWithCUMask(ctypes.byref(handle), len(words), arr)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: hipError {rc}")
    return torch.cuda.ExternalStream(handle.value, device=device)


def partition(device, teacher_per_xcd=16, layout="cu", xcds=8, cus_per_xcd=32):
    """Two masked streams of `device`.  layout "cu": the teacher gets the first `teacher_per_xcd` CUs of EVERY XCD, the student
    the rest; layout "xcd": the teacher gets `teacher_per_xcd` WHOLE XCDs (the last ones), the student the others."""
    device = torch.device(device)
    if device.type != "cuda":
        return None
    if layout not in ("cu", "xcd"):
        raise ValueError(f"CU-mask layout {layout!r}: expected 'cu' or 'xcd'")
    limit = cus_per_xcd if layout == "cu" else xcds
    if not 0 < int(teacher_per_xcd) < limit:
        raise ValueError(f"teacher share {teacher_per_xcd} outside 1 .. {limit - 1} for layout {layout!r}")
    # the mask layout above is the 8-XCD x 32-CU part's (MI355X); another CU count (MI300X's 304, a partitioned mode) would drop or
    # mis-assign compute units and size cu_reserve from the wrong counts: no split there
    have = torch.cuda.get_device_properties(device).multi_processor_count
    if have != xcds * cus_per_xcd:
        raise RuntimeError(f"CU-masked split is laid out for {xcds} XCDs x {cus_per_xcd} CUs; this device reports {have} compute units")
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(teacher_per_xcd), layout)
    if key not in _CACHE:
        total = xcds * cus_per_xcd
        if layout == "cu":
            t_bits = [i for i in range(total) if i // xcds < teacher_per_xcd]
        else:
            t_bits = [i for i in range(total) if i % xcds >= xcds - teacher_per_xcd]
        t_set = set(t_bits)
        s_bits = [i for i in range(total) if i not in t_set]
        if not t_bits or not s_bits:
            raise ValueError("both partitions need compute units")
        _CACHE[key] = (masked_stream(s_bits, device), masked_stream(t_bits, device), len(s_bits), len(t_bits))
    return _CACHE[key]


def env_partition():
    """CCD_FWD_SPLIT="cu:16" / "xcd:4" (layout : teacher share) -> (layout, share) or None."""
    v = os.environ.get("CCD_FWD_SPLIT", "")
    if not v or v == "0":
        return None
    layout, _, n = v.partition(":")
    if layout not in ("cu", "xcd") or not n.isdigit() or int(n) < 1:
        raise ValueError(f"CCD_FWD_SPLIT={v!r}: expected 'cu:<CUs per XCD>' or 'xcd:<XCDs>'")
    return layout, int(n)
