#!/usr/bin/env python3
"""gemm_tn384.h against the 128-square TN kernel on the four ViT weight-gradient shapes (131072 rows): result check against
torch fp32, then warm timings (policy gemm_tn384 = 0 / 1).  usage: python tools/tn384_lab.py [rows]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops
from microbench import timeit

BF = torch.bfloat16


def phases(dev):
    """lab build (-DCCD_MLP_LAB, CCD_HIP_LIB=...): cycle totals per phase, waves 0 and 7 of the first 32 workgroups, fc1 shape."""
    R, P, Q = 131072, 1536, 384
    dY = torch.randn(R, P, device=dev).to(BF)
    X = torch.randn(R, Q, device=dev).to(BF)
    names = ["dma issue", "fragment reads -> first MFMA", "MFMA phase", "vmcnt wait", "barrier"]
    for lab in (4, 5):
        ops.policy_set("lab", lab)
        dW = torch.zeros(P, Q, device=dev)
        cs = torch.zeros(P, device=dev)
        for _ in range(3):
            ops.gemm_tn_colsum(dY, X, dW, cs, splits=0)
        torch.cuda.synchronize()
        ph = cs.view(torch.int64)[:64 * 6].view(32, 2, 6).double()
        for wi, wname in enumerate(("wave 0", "wave 7")):
            tot = ph[:, wi, :5].sum(1).mean().item()
            print(json.dumps({"lab": lab, "wave": wname, "ticks_per_wg": round(tot),
                              "share": {n: round(100 * ph[:, wi, i].mean().item() / tot, 1) for i, n in enumerate(names)}}), flush=True)
    ops.policy_set("lab", 0)


def pairs(dev, R=131072, E=384):
    """The two pairs of a block as ccd_gemm_tn_pair issues them, against the separate products."""
    for name, (P1, Q1), (P2, Q2) in [("mlp", (E, 4 * E), (4 * E, E)), ("attn", (E, E), (3 * E, E))]:
        a1 = torch.randn(R, P1, device=dev).to(BF); b1 = torch.randn(R, Q1, device=dev).to(BF)
        a2 = torch.randn(R, P2, device=dev).to(BF); b2 = torch.randn(R, Q2, device=dev).to(BF)
        c1 = torch.zeros(P1, Q1, device=dev); c2 = torch.zeros(P2, Q2, device=dev)
        flops = 2.0 * R * (P1 * Q1 + P2 * Q2)
        with ops.policy(lab=1):
            ms1 = timeit(lambda: ops.gemm_tn_pair(a1, b1, c1, a2, b2, c2), iters=20)
        print(json.dumps({"pair": name, "main_loop_ms": round(ms1, 4)}), flush=True)
        for what, fn, pol in (("pair, split-K workspace + reduction (round 4)", lambda: ops.gemm_tn_pair(a1, b1, c1, a2, b2, c2), {}),
                              ("pair, fp32 atomics", lambda: ops.gemm_tn_pair(a1, b1, c1, a2, b2, c2, workspace=False), {}),
                              ("two calls, defaults", lambda: (ops.gemm_tn(a1, b1, c1), ops.gemm_tn(a2, b2, c2)), {}),
                              ("two calls, 128-square", lambda: (ops.gemm_tn(a1, b1, c1), ops.gemm_tn(a2, b2, c2)), dict(gemm_tn384=0))):
            with ops.policy(**pol):
                ms = timeit(fn, iters=20)
            print(json.dumps({"pair": name, "how": what, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1)}), flush=True)


def main():
    dev = torch.device("cuda:0")
    if os.environ.get("TN3_PHASES"):
        return phases(dev)
    if os.environ.get("TN3_PAIRS"):
        return pairs(dev)
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    E = 384
    torch.manual_seed(0)
    for name, P, Q in [("qkv", 3 * E, E), ("proj", E, E), ("fc1", 4 * E, E), ("fc2", E, 4 * E)]:
        dY = torch.randn(R, P, device=dev).to(BF)
        X = torch.randn(R, Q, device=dev).to(BF)
        ref = (dY[:8192].float().t() @ X[:8192].float())
        for pol in (0, 1):
            ops.policy_set("gemm_tn384", pol)
            dW = torch.zeros(P, Q, device=dev)
            ops.gemm_tn(dY[:8192], X[:8192], dW)
            err = float((dW - ref).abs().max())
            dW = torch.zeros(P, Q, device=dev)
            ms = timeit(lambda: ops.gemm_tn(dY, X, dW), iters=20)
            print(json.dumps({"shape": f"{name} {P}x{Q}x{R}", "tn384": pol, "max_err_8192": round(err, 4), "ms": round(ms, 4),
                              "tflops": round(2.0 * R * P * Q / ms / 1e9, 1),
                              "algorithmic_gbs": round(2.0 * R * (P + Q) / ms / 1e6, 1)}), flush=True)
        for lab, what in ((1, "main loop only"), (2, "one stage + epilogue")):
            ops.policy_set("lab", lab)
            ms = timeit(lambda: ops.gemm_tn(dY, X, dW), iters=20)
            print(json.dumps({"shape": name, "lab": what, "ms": round(ms, 4)}), flush=True)
        ops.policy_set("lab", 0)
    ops.policy_set("gemm_tn384", 1)


if __name__ == "__main__":
    main()
