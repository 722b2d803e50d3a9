#!/usr/bin/env python3
"""Lab: does a forward pass over HALF the rows at a time run faster per row?  (Every intermediate of a layer - qkv 302 MB, the
attention output, the residual stream - is then small enough to stay in the 256-MB Infinity Cache between its producer and its
consumer.)  Times the teacher's backbone pass (no saved state, no DropPath) over the 512 views of a 256-image batch in 1, 2, 4, 8
pieces, HIP events on the current stream.
usage: python tools/row_split_lab.py [--batch 256]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from ccd_amd import pretrain
from ccd_amd.synthetic import make_batch
from mlp_lab import timeit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    student, teacher = pretrain.build_networks(arch="vit_small", out_dim=65536, drop_path_rate=0.1, device=dev)
    teacher.eval()
    images, _, _ = make_batch(a.batch, seed=1, device=dev)
    views = torch.cat([images[:, 1], images[:, 2]]).contiguous()
    bb = teacher.backbone
    with torch.no_grad():
        for pieces in (1, 2, 4, 8, 1, 2, 4):
            n = views.shape[0] // pieces

            def run():
                return [bb.tokens_and_taps(views[i * n:(i + 1) * n], need_taps=False)[0] for i in range(pieces)]
            ms = timeit(run)
            print(json.dumps({"pass": "teacher backbone forward", "views": views.shape[0], "pieces": pieces, "ms": round(ms, 3)}), flush=True)


if __name__ == "__main__":
    main()
