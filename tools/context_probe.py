#!/usr/bin/env python3
"""Why does a streaming kernel take 2x longer inside the step than alone?  bn_relu_fwd on [1 Mi, 128] bf16 (536 MB of traffic),
timed with HIP events around the kernel only, in four contexts: the same buffers again and again; rotating over buffers that do not
fit the 256-MiB Infinity Cache; behind a heavy MFMA kernel; behind a kernel that has just written its input.
usage: python tools/context_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops

dev = torch.device("cuda:0")
BF = torch.bfloat16
rows, C = 1 << 20, 128
mr = torch.cat([torch.zeros(C, device=dev), torch.ones(C, device=dev)])
ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
xs = [torch.randn(rows, C, device=dev).to(BF) for _ in range(6)]
ys = [torch.empty_like(xs[0]) for _ in range(6)]
# a heavy MFMA kernel: the fused MLP on 131072 rows
R, E, H = 131072, 384, 1536
g = torch.Generator().manual_seed(0)
yy = torch.randn(R, E, generator=g).to(BF).to(dev)
w1 = (torch.randn(H, E, generator=g) * 0.05).to(BF).to(dev); w2 = (torch.randn(E, H, generator=g) * 0.03).to(BF).to(dev)
b1, b2 = torch.zeros(H, device=dev), torch.zeros(E, device=dev)
resid = torch.randn(R, E, generator=g).to(dev)
gam, bet = torch.ones(E, device=dev), torch.zeros(E, device=dev)


def heavy():
    ops.mlp_fused(yy, w1, b1, w2, b2, resid=resid, rowscale=None, rows_per_sample=256, gamma=gam, beta=bet, eps=1e-6, store_u=False)


def timed(before, pick, n=30):
    for _ in range(10):
        before(0); ops.bn_relu_fwd(xs[0], mr, ga, be, ys[0])
    torch.cuda.synchronize()
    tot = 0.0
    evs = []
    for k in range(n):
        i = pick(k)
        before(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.bn_relu_fwd(xs[i], mr, ga, be, ys[i])
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / n


res = {
    "same buffers": timed(lambda i: None, lambda k: 0),
    "rotating over 6 pairs (3.2 GB)": timed(lambda i: None, lambda k: k % 6),
    "behind a heavy MFMA kernel, same buffers": timed(lambda i: heavy(), lambda k: 0),
    "behind a heavy MFMA kernel, rotating": timed(lambda i: heavy(), lambda k: k % 6),
    "input just written by a streaming kernel, rotating": timed(lambda i: torch.clamp_min(xs[(i + 3) % 6], -100.0, out=xs[i]), lambda k: k % 6),
}
for k, v in res.items():
    print(json.dumps({"context": k, "bn_relu_fwd_ms": round(v, 4), "GB/s": round(rows * C * 4.0 / v / 1e6, 1)}), flush=True)
