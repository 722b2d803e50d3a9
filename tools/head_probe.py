#!/usr/bin/env python3
"""Lab timing of the DINO head's last layer (rows x 65536 x 256, bf16 out) with the tiled kernels, and of its data gradient (HIP
events, one MI355X).  usage: python tools/head_probe.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccd_amd import ops
from tools.mlp_lab import timeit
dev = torch.device("cuda:0"); BF = torch.bfloat16
M, N, K = 3328, 65536, 256
a = torch.randn(M, K, device=dev).to(BF); b = (torch.randn(N, K, device=dev) * 0.05).to(BF)
out = torch.empty(M, N, device=dev, dtype=BF)
for name, pol in (("default", {}), ("gemm256 forced", dict(gemm_256_min_m=1)), ("gemm256 deep", dict(gemm_256_min_m=1, gemm_256_deep=1))):
    with ops.policy(**pol):
        ms = timeit(lambda: ops.gemm_nt(a, b, out=out))
    print(json.dumps({"head last layer": name, "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1), "write_gbs": round(M * N * 2 / ms / 1e6, 1)}), flush=True)
# the data gradient: d_logits [M, 65536] . W [65536, 256] -> [M, 256] (K = 65536)
g = torch.randn(M, N, device=dev).to(BF); wt = (torch.randn(K, N, device=dev) * 0.05).to(BF)
ms = timeit(lambda: ops.gemm_nt(g, wt, epilogue=ops.EPI_F32))
print(json.dumps({"head dgrad": "default", "ms": round(ms, 4)}), flush=True)
