#!/usr/bin/env python3
"""Lab yardstick: the library's plain NT product (ccd_gemm_nt, bf16 in / bf16 out) against the vendor library's (torch.matmul -> hipBLASLt) on
the shapes of the generic path (config #4: E = 512 / 768) and of the headline step.  The vendor GEMM is NOT part of the product - it only says
what the part delivers on a shape.  HIP events, one MI355X.   usage: python tools/gemm_yardstick.py"""
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
SHAPES = [  # (M, N, K, what)
    (65536, 2048, 512, "vit_base fc1"), (65536, 512, 2048, "vit_base fc2"), (65536, 1536, 512, "vit_base qkv"), (65536, 512, 512, "vit_base proj"),
    (65536, 3072, 768, "768/12 fc1"), (65536, 768, 3072, "768/12 fc2"), (65536, 2304, 768, "768/12 qkv"), (65536, 768, 768, "768/12 proj"),
    (131072, 1536, 384, "vit_small fc2 data gradient (the gelu' product's shape)"), (131072, 1152, 384, "vit_small qkv"),
]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for M, N, K, what in SHAPES:
        a = torch.randn(M, K, generator=g).to(BF).to(dev)
        b = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
        out = torch.empty(M, N, dtype=BF, device=dev)
        ms_own = timeit(lambda: ops.gemm_nt(a, b, out=out))
        with ops.policy(gemm_256_deep=1):
            ms_deep = timeit(lambda: ops.gemm_nt(a, b, out=out))
        bt = b.t()
        ms_lib = timeit(lambda: torch.matmul(a, bt, out=out))
        fl = 2.0 * M * N * K
        print(json.dumps({"shape": [M, N, K], "what": what, "ccd_gemm_nt_ms": round(ms_own, 4), "ccd_tflops": round(fl / ms_own / 1e9, 1), "ccd_deep_ms": round(ms_deep, 4), "ccd_deep_tflops": round(fl / ms_deep / 1e9, 1),
                          "vendor_ms": round(ms_lib, 4), "vendor_tflops": round(fl / ms_lib / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
