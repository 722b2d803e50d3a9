#!/usr/bin/env python3
"""Lab timing of ccd_proj_mlp_fused (mlp_fused.h, PROJ) against the two launches it replaces (HIP events, one MI355X).
usage: python tools/proj_mlp_lab.py [--rows 131072]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from ccd_amd import ops
from mlp_lab import timeit

BF = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R, E, H = a.rows, 384, 1536
    g = torch.Generator().manual_seed(0)
    mk = lambda *s, dt=BF, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dt).to(dev)
    att, wp, w1, w2 = mk(R, E), mk(E, E, sc=0.05), mk(H, E, sc=0.05), mk(E, H, sc=0.03)
    bp, b1, b2 = mk(E, dt=torch.float32, sc=0.1), mk(H, dt=torch.float32, sc=0.1), mk(E, dt=torch.float32, sc=0.1)
    resid = mk(R, E, dt=torch.float32)
    one, zero = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    flops = 4.0 * R * E * H + 2.0 * R * E * E

    def fused(save):
        return ops.proj_mlp_fused(att, wp, bp, resid=resid, rowscale1=None, gamma2=one, beta2=zero, w1=w1, b1=b1, w2=w2, b2=b2,
                                  rowscale2=None, rows_per_sample=256, gamma=one, beta=zero, eps=1e-6, save=save)

    def pair(save):
        xm, y2, _, _ = ops.gemm_nt_resid_ln(att, wp, bias=bp, resid=resid, rowscale=None, rows_per_sample=256, gamma=one, beta=zero, eps=1e-6)
        return ops.mlp_fused(y2, w1, b1, w2, b2, resid=xm, rowscale=None, rows_per_sample=256, gamma=one, beta=zero, eps=1e-6, store_u=save)

    for save in (False, True):
        for name, fn in (("proj_mlp_fused", fused), ("resid_ln + mlp_fused", pair)):
            ms = timeit(lambda: fn(save))
            print(json.dumps({"kernel": name + ("+saved" if save else ""), "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1)}), flush=True)
    if os.environ.get("MLP_PHASES"):      # lab build (-DCCD_MLP_LAB via CCD_HIP_LIB): cycle totals of wave 0 per phase
        names = ["row loads", "acquire wait+barrier", "dma issue", "P1 plain", "P1 + gelu", "gelu tail / u store", "P2 + projection", "row passes"]
        for save in (False, True):
            out = fused(save)
            torch.cuda.synchronize()
            ph = out[2].view(torch.int64)[:256 * 8].view(256, 8).double()
            tot = ph.sum(1).mean().item()
            print(json.dumps({"kernel": "phases" + ("+saved" if save else ""), "cycles_per_wg": round(tot),
                              "share": {n: round(100 * ph[:, i].mean().item() / tot, 1) for i, n in enumerate(names)}}), flush=True)


if __name__ == "__main__":
    main()
