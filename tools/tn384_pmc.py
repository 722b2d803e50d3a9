#!/usr/bin/env python3
"""A few launches of gemm_tn384.h on the fc1 weight-gradient shape for the PMC passes of tools/tn384_pmc.sh.
usage: python tools/tn384_pmc.py [lab]   (lab 1 = main loop only)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops

dev = torch.device("cuda:0")
R, P, Q = 131072, 1536, 384
dY = torch.randn(R, P, device=dev).to(torch.bfloat16)
X = torch.randn(R, Q, device=dev).to(torch.bfloat16)
dW = torch.zeros(P, Q, device=dev)
ops.policy_set("lab", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for _ in range(6):
    ops.gemm_tn(dY, X, dW)
torch.cuda.synchronize()
