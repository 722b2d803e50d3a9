#!/usr/bin/env python3
"""Weight-gradient (TN) GEMM timing against the number of contraction slices (ccd_gemm_tn `splits`):
slices whose count is a multiple of the 8 XCDs do not straddle two L2s.  usage: python tools/tn_sweep.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops
from microbench import timeit

BF = torch.bfloat16


def main():
    dev = torch.device("cuda:0")
    R, E = 131072, 384
    for name, P, Q, cands in [("qkv", 3 * E, E, (0, 8, 16, 24)), ("proj", E, E, (0, 24, 32, 48, 64)),
                              ("fc1", 4 * E, E, (0, 8, 16, 24)), ("fc2", E, 4 * E, (0, 8, 16, 24))]:
        dY = torch.randn(R, P, device=dev).to(BF)
        X = torch.randn(R, Q, device=dev).to(BF)
        dW = torch.zeros(P, Q, device=dev)
        for s in cands:
            ms = timeit(lambda: ops.gemm_tn(dY, X, dW, splits=s), iters=10)
            print(json.dumps({"shape": f"{name} {P}x{Q}x{R}", "splits": s or "auto", "ms": round(ms, 4),
                              "tflops": round(2.0 * R * P * Q / ms / 1e9, 1),
                              "algorithmic_gbs": round(2.0 * R * (P + Q) / ms / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
