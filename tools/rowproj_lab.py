#!/usr/bin/env python3
"""Lab timing of the K = 384 / 512 bf16 projections: rowproj.h (activation rows resident in registers) against the tiled
kernels (gemm256.h / gemm.h), and of the attention backward with / without the fused qkv-bias gradient (HIP events, one MI355X).
usage: python tools/rowproj_lab.py [--rows 131072]"""
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
    ap.add_argument("--rows", type=int, default=131072)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    M = a.rows
    for K, N, what in () if os.environ.get("ATTN_ONLY") else ((384, 1152, "qkv"), (384, 384, "proj data gradient"), (384, 1536, "fc1-shaped"), (512, 1536, "qkv E=512"),
                       (512, 512, "proj data gradient E=512")):
        rows = M if K == 384 else M // 2
        x = torch.randn(rows, K, generator=g).to(BF).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.empty(rows, N, dtype=BF, device=dev)
        for rp in (0, 1):
            with ops.policy(rowproj=rp):
                ms = timeit(lambda: ops.gemm_nt(x, w, bias=bias, out=out))
            print(json.dumps({"product": what, "M": rows, "N": N, "K": K, "rowproj": rp, "ms": round(ms, 4),
                              "tflops": round(2.0 * rows * N * K / ms / 1e9, 1),
                              "algorithmic_gbs": round(2.0 * (rows * K + N * K + rows * N) / ms / 1e6, 1)}), flush=True)
    heads, E, views = 6, 384, M // 256
    qkv = torch.randn(views, 256, 3 * E, generator=g).to(BF).to(dev)
    d_out = torch.randn(views, 256, E, generator=g).to(BF).to(dev)
    out, lse = ops.attention_fwd(qkv, heads, 0.125)
    db = torch.zeros(3 * E, device=dev)
    dcs, eye = torch.zeros(E, device=dev), torch.eye(E, device=dev)
    dq = ops.attention_bwd(qkv, out, d_out, lse, heads, 0.125).view(-1, 3 * E)
    cs = torch.zeros(3 * E, device=dev)
    ms2 = timeit(lambda: ops.colsum_bf16(dq, cs))
    ms0 = timeit(lambda: ops.attention_bwd(qkv, out, d_out, lse, heads, 0.125))
    ms1 = timeit(lambda: ops.attention_bwd(qkv, out, d_out, lse, heads, 0.125, d_bias=db, dout_colsum=dcs, dout_colsum_mat=eye))
    print(json.dumps({"kernel": "attention_bwd", "views": views, "ms_plain": round(ms0, 4),
                      "ms_with_qkv_bias_gradient": round(ms1, 4), "ms_separate_colsum_bf16": round(ms2, 4)}), flush=True)


if __name__ == "__main__":
    main()
