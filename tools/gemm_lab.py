#!/usr/bin/env python3
"""Tiny driver for profiling ONE GEMM shape under rocprofv3 (--pmc or --kernel-trace).
usage: python tools/gemm_lab.py {tn|nt} P Q K [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops

kind, P, Q, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
BF = torch.bfloat16
if kind == "tn":
    a = torch.randn(K, P, device=dev).to(BF); b = torch.randn(K, Q, device=dev).to(BF)
    out = torch.zeros(P, Q, device=dev)
    fn = lambda: ops.gemm_tn(a, b, out)
else:
    a = torch.randn(P, K, device=dev).to(BF); b = torch.randn(Q, K, device=dev).to(BF)
    out = torch.empty(P, Q, device=dev, dtype=BF)
    mf = int(os.environ.get("LAB_MFAST", "0"))
    stamps = torch.zeros(8 * 1024 * 2, device=dev) if mf & 64 else None          # 8 u64 per workgroup (lab build)
    fn = lambda: ops.gemm_nt(a, b, out=out, m_fastest=mf, colsum=stamps)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"{kind} {P}x{Q}x{K}: {ms:.4f} ms  {2.0 * P * Q * K / ms / 1e9:.1f} TFLOP/s")

if kind == "nt" and mf & 64:
    st = stamps.view(torch.int64).view(-1, 8)[:512].double().cpu()
    names = ["prologue", "barrier", "compute", "lds-store", "load-issue", "next-prefetch", "epi-stage", "epi-global"]
    if os.environ.get("CCD_GEMM_256", "1") != "0" and Q >= int(os.environ.get("CCD_GEMM_256_MIN_N", "384")):
        names = ["first-tile wait", "compute+dma issue", "vmcnt wait", "k barrier", "epi stage", "epi barriers", "epi rows", "-"]
        st = stamps.view(torch.int64).view(-1, 8)[:256].double().cpu()
    tot = st.sum(1).mean().item()
    print("per-workgroup cycles (wave 0, mean over workgroups), s_memtime ticks:")
    for i, n in enumerate(names):
        print(f"  {n:14s} {st[:, i].mean().item():12.0f}  {100 * st[:, i].mean().item() / tot:5.1f} %")
    print(f"  total          {tot:12.0f}")
