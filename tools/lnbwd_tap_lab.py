#!/usr/bin/env python3
"""Lab timing: a tap's LayerNorm backward folded into the qkv data-gradient product (ccd_gemm_nt_lnbwd_tap_g16) against the two
launches it replaces (ccd_gemm_nt_lnbwd_g16, then ccd_ln_bwd_g16 of the tap with the MLP tail).  HIP events, one MI355X.
usage: python tools/lnbwd_tap_lab.py [--rows 131072]"""
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
    R, E, K = a.rows, 384, 1152
    g = torch.Generator().manual_seed(0)
    mk = lambda *s, dt=BF, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dt).to(dev)
    d_qkv, w, x = mk(R, K), mk(E, K, sc=0.05), mk(R, E, dt=torch.float32)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()
    gamma, gamma_t = mk(E, dt=torch.float32), mk(E, dt=torch.float32)
    d_tap = mk(R, E)
    gs, gb = mk(R, E), torch.empty((R, E), dtype=BF, device=dev)
    acc = lambda: torch.zeros(E, device=dev)
    dg, db, dgt, dbt, dbias = acc(), acc(), acc(), acc(), acc()
    rowscale = torch.ones(R // 256, device=dev)

    def pair():
        ops.gemm_nt_lnbwd(d_qkv, w, x, mean, rstd, gamma, gs, dg, db, accumulate=True)
        ops.ln_bwd(d_tap, x, mean, rstd, gamma_t, gs, dgt, dbt, accumulate=True, gb=gb, rowscale=rowscale, rows_per_sample=256, dbias=dbias)

    def alone():
        ops.gemm_nt_lnbwd(d_qkv, w, x, mean, rstd, gamma, gs, dg, db, accumulate=True, gb=gb, rowscale=rowscale, rows_per_sample=256, dbias=dbias)

    def folded():
        ops.gemm_nt_lnbwd(d_qkv, w, x, mean, rstd, gamma, gs, dg, db, accumulate=True, gb=gb, rowscale=rowscale, rows_per_sample=256, dbias=dbias,
                          tap=(d_tap, gamma_t, dgt, dbt))

    for name, fn in (("product + separate tap ln_bwd", pair), ("product alone (with the tail)", alone), ("product with the tap folded in", folded),
                     ("product + separate tap ln_bwd", pair), ("product with the tap folded in", folded)):
        print(json.dumps({"what": name, "rows": R, "ms": round(timeit(fn), 4)}), flush=True)


if __name__ == "__main__":
    main()
