#!/usr/bin/env python3
"""Phase profile of the one-pass attention backward (lab build -DCCD_ATTB1_LAB via CCD_HIP_LIB): cycle totals of waves 0 and 7."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ccd_amd import ops, _lib
BF = torch.bfloat16
dev = torch.device("cuda:0")
views, heads, E = 512, 6, 384
g = torch.Generator().manual_seed(0)
qkv = torch.randn(views, 256, 3 * E, generator=g).to(BF).to(dev); d_out = torch.randn(views, 256, E, generator=g).to(BF).to(dev)
scale = 64 ** -0.5
out, lse = ops.attention_fwd(qkv, heads, scale)
names = ["block end -> barrier", "barrier waits (start, delta)", "fragments + delta + K^T", "score products", "softmax", "exchange write + dV dK", "step barrier", "dQ reads + products", "dQ store", "dK dV stores"]
n_ws = int(_lib.get().ccd_attention_bwd_ws_floats(views, heads))
ws = torch.zeros(n_ws, device=dev); d_qkv = torch.empty_like(qkv); delta = torch.empty(views * heads * 256, device=dev)
d_bias = torch.zeros(3 * E, device=dev); vec = torch.zeros(E, device=dev)
for _ in range(3):
    _lib.check(_lib.get().ccd_attention_bwd(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(d_out), _lib.ptr(lse), _lib.ptr(delta), _lib.ptr(d_qkv), views, heads, scale,
                                            _lib.ptr(d_bias), _lib.ptr(ws), _lib.ptr(vec), 0, 0, _lib.stream()), "attention_bwd")
torch.cuda.synchronize()
ph = ws.view(-1, E)[:256, :32].double()
for wi, nm in ((0, "wave 0"), (16, "wave 7")):
    tot = ph[:, wi:wi + 10].sum(1).mean().item()
    print(json.dumps({"wave": nm, "cycles_per_workgroup": round(tot), "cycles_per_block": round(tot / 12),
                      "share_percent": {n: round(100 * ph[:, wi + i].mean().item() / tot, 1) for i, n in enumerate(names)}}))
