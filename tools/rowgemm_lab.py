#!/usr/bin/env python3
"""Lab timing of the row-owner LayerNorm-backward product (rowgemm.h) against gemm_row384.h's (HIP events, one MI355X).
usage: python tools/rowgemm_lab.py [--rows 131584]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops
from tools.mlp_lab import timeit

BF = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[131584, 131072, 32768])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    E = 384
    g = torch.Generator().manual_seed(0)
    for M in a.rows:
        x = torch.randn(M, E, generator=g).to(dev)
        mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()
        gamma = torch.ones(E, device=dev)
        gbuf = torch.randn(M, E, generator=g).to(dev)
        dgam, dbet, dbias = torch.zeros(E, device=dev), torch.zeros(E, device=dev), torch.zeros(E, device=dev)
        gb = torch.zeros(M, E, dtype=BF, device=dev)
        rowscale = torch.ones(M // 256 + 1, device=dev)
        if os.environ.get("RG_PHASES"):      # lab build (-DCCD_MLP_LAB via CCD_HIP_LIB): cycle totals of wave 0 per phase
            names = ["tile start", "wait for my quarter", "barrier", "product", "pass A", "pass B", "flush", "-"]
            for K in (384, 1152, 1536):
              aa = torch.randn(M, K, generator=g).to(BF).to(dev)
              w = (torch.randn(E, K, generator=g) * 0.05).to(BF).to(dev)
              for lab in [0] + [int(v) for v in os.environ.get("RG_LAB", "").split(",") if v]:
                ops.policy_set("lab", lab)
                for tail in (False, True):
                    for _ in range(3):       # (the last launch is read: warm clocks and caches)
                        ops.gemm_nt_lnbwd(aa, w, x, mean, rstd, gamma, gbuf, dgam, dbet, accumulate=True, gb=gb if tail else None,
                                          rowscale=rowscale if tail else None, rows_per_sample=256, dbias=dbias if tail else None)
                    torch.cuda.synchronize()
                    ph = gbuf.view(-1).view(torch.int64)[:256 * 8].view(256, 8).double()
                    tot = ph.sum(1).mean().item()
                    print(json.dumps({"M": M, "K": K, "lab": lab, "tail": tail, "cycles_per_wg": round(tot),
                                      "share": {n: round(100 * ph[:, i].mean().item() / tot, 1) for i, n in enumerate(names)}}), flush=True)
            continue
        if not os.environ.get("RG_NO_RESID"):    # attn.proj + residual + LayerNorm (K = 384): gemm_row384.h (default) vs rowgemm.h
            aa = torch.randn(M, E, generator=g).to(BF).to(dev)
            w = (torch.randn(E, E, generator=g) * 0.05).to(BF).to(dev)
            bias, beta = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
            out = torch.empty(M, E, device=dev)
            for rowgemm in (1, 2):
                with ops.policy(rowgemm=rowgemm):
                    ms = timeit(lambda: ops.gemm_nt_resid_ln(aa, w, bias=bias, resid=x, rowscale=rowscale, rows_per_sample=256,
                                                             gamma=gamma, beta=beta, eps=1e-6, out=out))
                print(json.dumps({"kernel": "resid_ln", "M": M, "K": E, "rowgemm": rowgemm, "ms": round(ms, 4),
                                  "tflops": round(2.0 * M * E * E / ms / 1e9, 1),
                                  "algorithmic_gbs": round(M * E * 12.0 / ms / 1e6, 1)}), flush=True)
        for K in (384, 1152, 1536):
            aa = torch.randn(M, K, generator=g).to(BF).to(dev)
            w = (torch.randn(E, K, generator=g) * 0.05).to(BF).to(dev)
            for rowgemm, lab, rg8 in [(1, 0, 1), (1, 0, 0), (0, 0, 0)] + [(1, int(v), 1) for v in os.environ.get("RG_LAB", "").split(",") if v]:
                for tail in (False, True):
                    for acc in (True,) if os.environ.get("RG_QUICK") else (True, False):
                        with ops.policy(rowgemm=rowgemm, lab=lab, rowgemm_adma=rg8):
                            ms = timeit(lambda: ops.gemm_nt_lnbwd(aa, w, x, mean, rstd, gamma, gbuf, dgam, dbet, accumulate=acc,
                                                                  gb=gb if tail else None, rowscale=rowscale if tail else None,
                                                                  rows_per_sample=256, dbias=dbias if tail else None))
                        nbytes = M * (2.0 * K + E * (4 + 4 + (4 if acc else 0) + (2 if tail else 0)))
                        print(json.dumps({"M": M, "K": K, "rowgemm": rowgemm, "adma": rg8, "lab": lab, "tail": tail, "acc": acc, "ms": round(ms, 4),
                                          "tflops": round(2.0 * M * E * K / ms / 1e9, 1), "algorithmic_gbs": round(nbytes / ms / 1e6, 1)}),
                              flush=True)


if __name__ == "__main__":
    main()
