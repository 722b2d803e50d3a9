"""Is the weight-gradient pair with a workspace the same bits on every run? (GPU)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kernel_checks as kc
from ccd_amd import ops
BF = torch.bfloat16
dev = torch.device("cuda:0")
for Mc, s1, s2 in ((8192, (384, 1536), (1536, 384)), (8224, (384, 1536), (1536, 384)), (131072, (384, 1536), (1536, 384)), (131072, (384, 384), (384, 1152)), (8224, (384, 384), (384, 1152))):
    g = torch.Generator().manual_seed(8)
    bases, args, wants = [], [], []
    for P, Q in (s1, s2):
        a = kc.rnd((Mc, P), g).to(BF); b = kc.rnd((Mc, Q), g).to(BF)
        bases.append(kc.rnd((P, Q), g)); args.append((a.to(dev), b.to(dev)))
    for ws in (True, False):
        res = []
        for rep in range(4):
            outs = [x.clone().to(dev) for x in bases]
            ops.gemm_tn_pair(args[0][0], args[0][1], outs[0], args[1][0], args[1][1], outs[1], workspace=ws)
            torch.cuda.synchronize()
            res.append([o.clone() for o in outs])
        d = [max((res[r][i] - res[0][i]).abs().max().item() for r in range(1, 4)) for i in (0, 1)]
        n = [max(int(((res[r][i] - res[0][i]) != 0).sum()) for r in range(1, 4)) for i in (0, 1)]
        print(Mc, s1, s2, "workspace" if ws else "atomics", "max diff between runs", d, "elements differing", n, flush=True)
