#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU (HIP events): torch's own copy / relu / sum over tensors below and above the
256-MiB Infinity Cache - the yardstick for the elementwise kernels of the step.  usage: python tools/stream_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.mlp_lab import timeit

dev = torch.device("cuda:0")
for mb in (32, 128, 268, 536, 1072):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("copy", lambda: y.copy_(x), 4.0 * n), ("relu", lambda: torch.relu(x, out=y) if False else torch.clamp_min(x, 0, out=y), 4.0 * n),
                             ("sum", lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32), 2.0 * n)):
        ms = timeit(fn)
        print(json.dumps({"op": name, "tensor_MB": mb, "ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1)}), flush=True)

# the step's own streaming kernels on the same yardstick (CCD_STREAM_OPS=1)
if os.environ.get("CCD_STREAM_OPS"):
    from ccd_amd import ops
    for rows, C in ((1 << 20, 128), (1 << 18, 128), (1 << 16, 64)):
        x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        dy = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        y = torch.empty_like(x)
        mr = torch.cat([torch.zeros(C, device=dev), torch.ones(C, device=dev)])
        ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        red, zl = torch.zeros(2 * C, device=dev), torch.zeros(2 * C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        nb = rows * C * 2.0
        for name, fn, nbytes in (("bn_relu_fwd", lambda: ops.bn_relu_fwd(x, mr, ga, be, y), 2 * nb),
                                 ("bn_relu_bwd_reduce", lambda: ops.bn_relu_bwd_reduce(dy, x, mr, ga, be, red), 2 * nb),
                                 ("bn_relu_bwd_apply", lambda: ops.bn_relu_bwd_apply(dy, x, mr, ga, be, red, float(rows), zl, dg, db, y), 3 * nb),
                                 ("torch relu", lambda: torch.clamp_min(x, 0, out=y), 2 * nb)):
            ms = timeit(fn)
            print(json.dumps({"op": name, "rows": rows, "C": C, "ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1)}), flush=True)
