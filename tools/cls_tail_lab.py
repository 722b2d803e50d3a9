"""Timings of the fused classifier tail (kernels/cls_tail.h) against the conv.h kernels it replaces, on the step's own shape.

    python tools/cls_tail_lab.py [--images 512] [--reps 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lab", type=int, default=0)
    ap.add_argument("--fused-only", action="store_true")
    a = ap.parse_args()
    from ccd_amd import ops, seghead as sh
    dev = torch.device("cuda", 0)
    ops.policy_set("lab", a.lab)
    C, H, W, n = 128, 32, 128, a.images
    P = n * H * W
    g = torch.Generator().manual_seed(0)
    y = (torch.randn((P, C), generator=g) * 1.3).to(torch.bfloat16).to(dev)
    yf = y[: 1 << 16].float()
    mean_rstd = torch.cat([yf.mean(0), torch.rsqrt(yf.var(0) + 1e-5)]).contiguous()
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) * 0.2
    w, bias = torch.randn((2, C, 3, 3), device=dev) * 0.05, torch.zeros(2, device=dev)
    dl = torch.randn((n, 2, H, W), device=dev) / 64
    red, db = torch.zeros(2 * C, device=dev), torch.zeros(2, device=dev)
    dgamma, dbeta, dw, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros_like(w), torch.zeros(C, device=dev)
    dy = torch.empty_like(y)
    ybytes = P * C * 2
    out = {"lab": a.lab, "images": n, "y_MB": round(ybytes / 1e6, 1)}
    t = timed(lambda: ops.cls_tail_fwd(y, mean_rstd, gamma, beta, w, bias, n, H, W), a.reps)
    out["fwd_us"], out["fwd_TBps_of_y"] = round(t, 1), round(ybytes / t / 1e6, 2)
    t = timed(lambda: ops.cls_tail_bwd_reduce(dl, y, mean_rstd, gamma, beta, w, red, db, n, H, W), a.reps)
    out["bwd_reduce_us"], out["bwd_reduce_TBps_of_y"] = round(t, 1), round(ybytes / t / 1e6, 2)
    t = timed(lambda: ops.cls_tail_bwd_apply(dl, y, mean_rstd, gamma, beta, w, red, float(P), red, dgamma, dbeta, dw, dbt, dy, n, H, W), a.reps)
    out["bwd_apply_us"], out["bwd_apply_TBps_of_2y"] = round(t, 1), round(2 * ybytes / t / 1e6, 2)
    if a.fused_only:
        print(json.dumps(out), flush=True)
        return
    # the unfused chain
    abuf = torch.empty_like(y)
    t1 = timed(lambda: ops.bn_relu_fwd(y, mean_rstd, gamma, beta, abuf), a.reps)
    t2 = timed(lambda: sh.cls_forward(abuf, w, bias, n, H, W), a.reps)
    out["unfused_fwd_us"] = [round(t1, 1), round(t2, 1)]
    dwz, dbz = torch.zeros_like(w), torch.zeros(2, device=dev)
    t3 = timed(lambda: sh.cls_backward(dl, abuf, w, dwz, dbz, n, H, W), max(3, a.reps // 4))
    dx = sh.cls_backward(dl, abuf, w, dwz, dbz, n, H, W)
    t4 = timed(lambda: ops.bn_relu_bwd_reduce(dx, y, mean_rstd, gamma, beta, red), a.reps)
    t5 = timed(lambda: ops.bn_relu_bwd_apply(dx, y, mean_rstd, gamma, beta, red, float(P), red, dgamma, dbeta, dy), a.reps)
    t6 = timed(lambda: ops.colsum_bf16(dy, dbt), a.reps)
    out["unfused_bwd_us"] = [round(x, 1) for x in (t3, t4, t5, t6)]
    out["fused_total_us"] = round(out["fwd_us"] + out["bwd_reduce_us"] + out["bwd_apply_us"], 1)
    out["unfused_total_us"] = round(t1 + t2 + t3 + t4 + t5 + t6, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
