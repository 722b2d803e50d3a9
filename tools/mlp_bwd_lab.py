#!/usr/bin/env python3
"""Lab timing: the MLP branch's data-gradient chain in one launch (ccd_mlp_bwd_fused, mlp_bwd.h) against the two launches it
replaces (ccd_gemm_nt with the gelu'(u) epilogue writing du + gelu(u), then ccd_gemm_nt_lnbwd_g16 reading du back).  HIP events, one MI355X.
usage: python tools/mlp_bwd_lab.py [--rows 131072]"""
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
    ap.add_argument("--E", type=int, default=384)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R, E = a.rows, a.E
    H = 4 * E
    g = torch.Generator().manual_seed(0)
    mk = lambda *s, dt=BF, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dt).to(dev)
    gb, u = mk(R, E, sc=0.5), mk(R, H, sc=1.5)
    w2t, w1t = mk(H, E, sc=0.06), mk(E, H, sc=0.08)
    x = mk(R, E, dt=torch.float32)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()
    gamma = mk(E, dt=torch.float32)
    gs, gbo = mk(R, E), torch.empty((R, E), dtype=BF, device=dev)
    acc = lambda n=E: torch.zeros(n, device=dev)
    dg, db, dbias, db1 = acc(), acc(), acc(), acc(H)
    rowscale = torch.ones(R // 256, device=dev)
    gact = torch.empty((R, H), dtype=BF, device=dev)

    def dgelu():
        return ops.gemm_nt(gb, w2t, epilogue=ops.EPI_DGELU, aux=u, out2=gact, colsum=db1)

    du0 = dgelu()

    def lnbwd():
        ops.gemm_nt_lnbwd(du0, w1t, x, mean, rstd, gamma, gs, dg, db, accumulate=True, gb=gbo, rowscale=rowscale, rows_per_sample=256, dbias=dbias)

    def pair():
        du = dgelu()
        ops.gemm_nt_lnbwd(du, w1t, x, mean, rstd, gamma, gs, dg, db, accumulate=True, gb=gbo, rowscale=rowscale, rows_per_sample=256, dbias=dbias)

    def fused():
        ops.mlp_bwd_fused(gb, w2t, w1t, u, db1=db1, x=x, mean=mean, rstd=rstd, gamma=gamma, g=gs, dgamma=dg, dbeta=db, gb_out=gbo,
                          rowscale=rowscale, rows_per_sample=256, dbias=dbias)

    for name, fn in (("gelu'(u) product (du + gelu(u) out)", dgelu), ("LayerNorm-backward product (du in)", lnbwd), ("the two launches", pair),
                     ("fused", fused), ("the two launches", pair), ("fused", fused)):
        print(json.dumps({"what": name, "rows": R, "E": E, "ms": round(timeit(fn), 4)}), flush=True)


if __name__ == "__main__":
    main()
