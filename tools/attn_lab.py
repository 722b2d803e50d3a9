#!/usr/bin/env python3
"""Lab timing of the attention kernels (HIP events, one MI355X): the policy key `attn_skew` = skew of waves 4..7 in the backward
kernels (x 64 cycles).  usage: python tools/attn_lab.py [--views 512]"""
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
    ap.add_argument("--views", type=int, default=512)
    ap.add_argument("--skews", type=int, nargs="+", default=[0, 2, 4, 6, 8, 12, 16, 24])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    heads, E = 6, 384
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(a.views, 256, 3 * E, generator=g).to(BF).to(dev)
    d_out = torch.randn(a.views, 256, E, generator=g).to(BF).to(dev)
    scale = 64 ** -0.5
    out, lse = ops.attention_fwd(qkv, heads, scale)
    ms = timeit(lambda: ops.attention_fwd(qkv, heads, scale))
    print(json.dumps({"kernel": "attention_fwd", "views": a.views, "ms": round(ms, 4)}), flush=True)
    for tr in (1, 0):
        with ops.policy(attn_tr=tr, attn_skew=0):
            ms = timeit(lambda: ops.attention_bwd(qkv, out, d_out, lse, heads, scale))
        print(json.dumps({"kernel": "attention_bwd (dq + dkv)", "attn_tr": tr, "views": a.views, "ms": round(ms, 4)}), flush=True)
    d_bias = torch.zeros(3 * E, device=dev); dcs = torch.zeros(E, device=dev)
    for onepass in (1, 0):
        with ops.policy(attn_onepass=onepass):
            ms = timeit(lambda: ops.attention_bwd(qkv, out, d_out, lse, heads, scale))
            msb = timeit(lambda: ops.attention_bwd(qkv, out, d_out, lse, heads, scale, d_bias=d_bias, dout_colsum=dcs))
        print(json.dumps({"kernel": "attention_bwd one pass (round 4)" if onepass else "attention_bwd (dq + dkv_tr)", "views": a.views,
                          "ms": round(ms, 4), "ms_with_qkv_bias_gradient": round(msb, 4)}), flush=True)
    for lab, what in ((1, "no static priority"), (3, "priority on the older half")):
        with ops.policy(attn_onepass=1, lab=lab):
            ms = timeit(lambda: ops.attention_bwd(qkv, out, d_out, lse, heads, scale, d_bias=d_bias, dout_colsum=dcs))
        print(json.dumps({"kernel": "attention_bwd one pass", "lab": what, "ms_with_qkv_bias_gradient": round(ms, 4)}), flush=True)
    if os.environ.get("ATTN_LAB_SHORT"):
        return
    for lab, what in ((1, "no stores"), (2, "no loads"), (3, "no loads, no stores")):
        with ops.policy(attn_tr=1, attn_skew=0, lab=lab):
            ms = timeit(lambda: ops.attention_bwd(qkv, out, d_out, lse, heads, scale))
        print(json.dumps({"kernel": "attention_bwd (dq + dkv_tr)", "lab": what, "ms": round(ms, 4)}), flush=True)
    if os.environ.get("ATTN_LAB_SHORT"):
        return
    for skew in a.skews:
        with ops.policy(attn_skew=skew):
            ms = timeit(lambda: ops.attention_bwd(qkv, out, d_out, lse, heads, scale))
        print(json.dumps({"kernel": "attention_bwd (dq + dkv)", "views": a.views, "skew": skew, "ms": round(ms, 4)}), flush=True)


if __name__ == "__main__":
    main()
