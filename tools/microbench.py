#!/usr/bin/env python3
"""Per-kernel timing on the GPU box (HIP events on the current stream), printed as JSON lines.
usage: python tools/microbench.py [--out gpurun_out/microbench.jsonl]"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops

BF = torch.bfloat16
PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def timeit(fn, iters=20, warm=3, warm_seconds=0.5):
    import time
    t0, n = time.time(), 0
    while n < warm or time.time() - t0 < warm_seconds:      # clock ramp: see tools/mlp_lab.py
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--rows", type=int, default=131072)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R, E = a.rows, 384
    res = []

    def rec(name, ms, flops=None, bytes_=None):
        r = {"kernel": name, "ms": round(ms, 4)}
        if flops:
            r["tflops"] = round(flops / ms / 1e9, 1)
            r["frac_mfma_peak"] = round(flops / ms / 1e9 / PEAK_TF, 3)
        if bytes_:
            r["gbs"] = round(bytes_ / ms / 1e6, 1)
            r["frac_hbm_peak"] = round(bytes_ / ms / 1e6 / PEAK_GBS, 3)
        res.append(r)
        print(json.dumps(r), flush=True)

    x = torch.randn(R, E, device=dev)
    gamma, beta = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    rec("ln_fwd", timeit(lambda: ops.ln_fwd(x, gamma, beta)), bytes_=R * E * 6)
    y, mean, rstd = ops.ln_fwd(x, gamma, beta)
    g = torch.zeros_like(x); dg = torch.zeros(E, device=dev); db = torch.zeros(E, device=dev)
    rec("ln_bwd", timeit(lambda: ops.ln_bwd(y, x, mean, rstd, gamma, g, dg, db)), bytes_=R * E * 14)

    for name, N, K, epi in [("qkv", 3 * E, E, ops.EPI_BF16), ("proj_resid", E, E, ops.EPI_RESID),
                            ("fc1_gelu", 4 * E, E, ops.EPI_GELU), ("fc2_resid", E, 4 * E, ops.EPI_RESID)]:
        A = torch.randn(R, K, device=dev).to(BF)
        W = (torch.randn(N, K, device=dev) * 0.02).to(BF)
        bias = torch.zeros(N, device=dev)
        kw = dict(epilogue=epi, bias=bias)
        if epi == ops.EPI_RESID:
            kw["resid"] = torch.randn(R, N, device=dev)
            kw["out"] = torch.empty(R, N, device=dev)
        elif epi == ops.EPI_GELU:
            kw["out"] = torch.empty(R, N, device=dev, dtype=BF); kw["out2"] = torch.empty(R, N, device=dev, dtype=BF)
        else:
            kw["out"] = torch.empty(R, N, device=dev, dtype=BF)
        rec(f"gemm_nt_{name}_{R}x{N}x{K}", timeit(lambda: ops.gemm_nt(A, W, **kw)), flops=2.0 * R * N * K)
        dY = torch.randn(R, N, device=dev).to(BF)
        dW = torch.zeros(N, K, device=dev)
        rec(f"gemm_tn_{name}_{N}x{K}x{R}", timeit(lambda: ops.gemm_tn(dY, A, dW)), flops=2.0 * R * N * K)
    # square-ish reference point
    A = torch.randn(8192, 4096, device=dev).to(BF); W = torch.randn(8192, 4096, device=dev).to(BF)
    out = torch.empty(8192, 8192, device=dev, dtype=BF)
    rec("gemm_nt_8192x8192x4096", timeit(lambda: ops.gemm_nt(A, W, out=out), iters=5), flops=2.0 * 8192 * 8192 * 4096)
    # DINO last layer
    M2 = 3584
    A = torch.randn(M2, 256, device=dev).to(BF); W = torch.randn(65536, 256, device=dev).to(BF)
    out = torch.empty(M2, 65536, device=dev)
    rec("gemm_nt_logits_3584x65536x256", timeit(lambda: ops.gemm_nt(A, W, epilogue=ops.EPI_F32, out=out), iters=5),
        flops=2.0 * M2 * 65536 * 256)

    views, heads = R // 256, 6
    qkv = torch.randn(views, 256, 3 * E, device=dev).to(BF)
    fl = 4.0 * views * heads * 256 * 256 * 64
    rec("attention_fwd", timeit(lambda: ops.attention_fwd(qkv, heads, 0.125)), flops=fl)
    o, lse = ops.attention_fwd(qkv, heads, 0.125)
    do = torch.randn_like(o)
    rec("attention_bwd", timeit(lambda: ops.attention_bwd(qkv, o, do, lse, heads, 0.125)), flops=2.5 * fl)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            for r in res:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
