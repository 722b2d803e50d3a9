#!/usr/bin/env python3
"""Lab timing of the fused MLP kernel against the unfused pair it replaces (HIP events, one MI355X).
usage: python tools/mlp_lab.py [--rows 131072]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops

BF = torch.bfloat16


def timeit(fn, iters=20, warm=3, warm_seconds=0.5):
    """Mean ms per call.  Warm-up runs until `warm_seconds` of GPU work have passed: the first configuration of a lab loop
    measured 8 - 15 % slow with three warm-up calls (clock ramp) - an artefact that once read as a gain of whatever came second."""
    import time
    t0 = time.time()
    n = 0
    while n < warm or time.time() - t0 < warm_seconds:
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
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R, E, H = a.rows, 384, 1536
    g = torch.Generator().manual_seed(0)
    y = torch.randn(R, E, generator=g).to(BF).to(dev)
    w1 = (torch.randn(H, E, generator=g) * 0.05).to(BF).to(dev); w2 = (torch.randn(E, H, generator=g) * 0.03).to(BF).to(dev)
    b1, b2 = torch.randn(H, generator=g).to(dev) * 0.1, torch.randn(E, generator=g).to(dev) * 0.1
    resid = torch.randn(R, E, generator=g).to(dev)
    gamma, beta = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    flops = 4.0 * R * E * H

    def rec(name, ms, nbytes):
        print(json.dumps({"kernel": name, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1),
                          "algorithmic_gbs": round(nbytes / ms / 1e6, 1)}), flush=True)

    for store_u in (False, True):
        ms = timeit(lambda: ops.mlp_fused(y, w1, b1, w2, b2, resid=resid, rowscale=None, rows_per_sample=256, gamma=gamma,
                                          beta=beta, eps=1e-6, store_u=store_u))
        rec("mlp_fused" + ("+u" if store_u else ""), ms, R * E * 12.0 + (2.0 * R * H if store_u else 0.0))

    if os.environ.get("MLP_PHASES"):      # lab build (-DCCD_MLP_LAB via CCD_HIP_LIB): cycle totals of wave 0 per phase
        names = ["tile prologue", "acquire wait+barrier", "dma issue", "P1 plain", "P1 + gelu", "gelu tail / u store", "P2", "epilogue"]
        for store_u in (False, True):
            out = ops.mlp_fused(y, w1, b1, w2, b2, resid=resid, rowscale=None, rows_per_sample=256, gamma=gamma, beta=beta,
                                eps=1e-6, store_u=store_u)
            torch.cuda.synchronize()
            ph = out[2].view(torch.int64)[:256 * 8].view(256, 8).double()
            tot = ph.sum(1).mean().item()
            print(json.dumps({"kernel": "phases" + ("+u" if store_u else ""), "cycles_per_wg": round(tot),
                              "share": {n: round(100 * ph[:, i].mean().item() / tot, 1) for i, n in enumerate(names)}}), flush=True)
        return
    for lab in [int(x) for x in os.environ.get("MLP_LAB", "").split(",") if x]:
        with ops.policy(lab=lab):
            for store_u in (False, True):
                ms = timeit(lambda: ops.mlp_fused(y, w1, b1, w2, b2, resid=resid, rowscale=None, rows_per_sample=256, gamma=gamma,
                                                  beta=beta, eps=1e-6, store_u=store_u))
                print(json.dumps({"kernel": "mlp_fused lab=%d%s" % (lab, "+u" if store_u else ""), "ms": round(ms, 4)}), flush=True)
    # per-chunk slope and per-tile intercept: the same launch with other hidden sizes
    for Hs in [int(x) for x in os.environ.get("MLP_HS", "64,256,512,1024").split(",")]:
        w1s = (torch.randn(Hs, E, generator=g) * 0.05).to(BF).to(dev); w2s = (torch.randn(E, Hs, generator=g) * 0.03).to(BF).to(dev)
        b1s = torch.randn(Hs, generator=g).to(dev) * 0.1
        for store_u in (False, True):
            ms = timeit(lambda: ops.mlp_fused(y, w1s, b1s, w2s, b2, resid=resid, rowscale=None, rows_per_sample=256, gamma=gamma,
                                              beta=beta, eps=1e-6, store_u=store_u))
            print(json.dumps({"kernel": "mlp_fused H=%d%s" % (Hs, "+u" if store_u else ""), "ms": round(ms, 4)}), flush=True)
    if os.environ.get("MLP_LAB_ONLY_FUSED"):
        return

    def unfused(store_u):
        u, gact = ops.gemm_nt(y, w1, epilogue=ops.EPI_GELU, bias=b1, store_u=store_u)
        return ops.gemm_nt_resid_ln(gact, w2, bias=b2, resid=resid, rowscale=None, rows_per_sample=256, gamma=gamma,
                                    beta=beta, eps=1e-6)
    for store_u in (False, True):
        ms = timeit(lambda: unfused(store_u))
        rec("fc1_gelu + fc2_resid_ln" + ("+u" if store_u else ""), ms, R * E * 12.0 + 4.0 * R * H + (2.0 * R * H if store_u else 0.0))
    ms = timeit(lambda: ops.gemm_nt(y, w1, epilogue=ops.EPI_GELU, bias=b1, store_u=False))
    rec("fc1_gelu only", ms, R * E * 2.0 + 2.0 * R * H)


if __name__ == "__main__":
    main()
