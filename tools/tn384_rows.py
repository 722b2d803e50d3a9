#!/usr/bin/env python3
"""gemm_tn384.h main loop (policy lab = 1) against the size of the operands - L2-, MALL- and HBM-resident: the time per 32-row
stage does not depend on it (profiles/r02_tn384_lab.jsonl).  usage (GPU box): python tools/tn384_rows.py"""
import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from ccd_amd import ops
from microbench import timeit
dev = torch.device("cuda:0"); BF = torch.bfloat16
for R in (8192, 16384, 32768, 65536, 131072, 262144):
    P, Q = 1536, 384
    dY = torch.randn(R, P, device=dev).to(BF); X = torch.randn(R, Q, device=dev).to(BF); dW = torch.zeros(P, Q, device=dev)
    with ops.policy(lab=1):
        ms = timeit(lambda: ops.gemm_tn(dY, X, dW), iters=20)
    stages = R / 32 / 32
    print(json.dumps({"rows": R, "MB": round(R * (P + Q) * 2 / 1e6), "main_loop_ms": round(ms, 4), "stages": stages, "us_per_stage": round(1e3 * ms / stages, 3)}), flush=True)
