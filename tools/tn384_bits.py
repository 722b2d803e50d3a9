#!/usr/bin/env python3
"""gemm_tn384.h, fc1 weight-gradient shape (131072 rows): what the parts cost, by the kernel's lab switches (policy `lab`:
1 = no epilogue, 2 = one stage + epilogue, 8 = no LDS-DMA inside the loop, 16 = no barrier).  usage (GPU box): python tools/tn384_bits.py"""
import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from ccd_amd import ops
from microbench import timeit
dev = torch.device("cuda:0"); BF = torch.bfloat16
R, P, Q = 131072, 1536, 384
dY = torch.randn(R, P, device=dev).to(BF); X = torch.randn(R, Q, device=dev).to(BF); dW = torch.zeros(P, Q, device=dev)
for lab, what in ((0, "full"), (1, "main loop"), (2, "one stage + epilogue"), (9, "main loop without LDS-DMA"), (17, "main loop without barrier"), (25, "main loop without either")):
    with ops.policy(lab=lab):
        ms = timeit(lambda: ops.gemm_tn(dY, X, dW), iters=20)
    print(json.dumps({"lab": lab, "what": what, "ms": round(ms, 4), "tflops_equiv": round(2.0 * R * P * Q / ms / 1e9)}), flush=True)
