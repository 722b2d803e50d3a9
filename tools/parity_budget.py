#!/usr/bin/env python3
"""Where does the toy model's loss difference come from?  (the B = 2 smoke run differed from the oracle by
1.49e-3 while the B = 8 ViT-Small runs sit below 2e-4.)  One pretraining iteration of the 3-block E = 192 model on the HIP path and
on the CPU oracle for several batch sizes and seeds: loss deltas, the rms error of the student / teacher logits (the bf16 path's
per-row noise) and the number of selected rows M the DINO loss averages over.  If the delta is row noise it shrinks like
rms_logit_error / sqrt(M); a systematic error (a wrong LayerNorm path, the 512-wide loss) would not.
usage (GPU box): python tools/parity_budget.py > gpurun_out/parity_tiny_budget.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from ccd_amd import pretrain
from ccd_amd.loss.Dino_loss import DINOLoss
from ccd_amd.synthetic import make_batch
from oracle import ccd_oracle as O


def main():
    dev = torch.device("cuda:0")
    spec = O.Spec(embed_dim=192, depth=3, heads=3, taps=(1, 2, 3), out_dim=512, head_hidden=256, head_bottleneck=64,
                  norm_last_layer=False, seg_in=192)
    rows = []
    for B in (2, 4, 8, 16, 32):
        for seed in (11, 12, 13, 14):
            torch.manual_seed(3)
            np.random.seed(3)
            student, teacher = pretrain.build_networks(
                arch=None, out_dim=512, drop_path_rate=0.0, norm_last_layer=False, seg_channel=192,
                backbone_kwargs=dict(embed_dim=192, depth=3, num_heads=3, out_indices=[1, 2, 3]),
                head_kwargs=dict(hidden_dim=256, bottleneck_dim=64), device=dev)
            dino_loss = DINOLoss(512, 2, 0.04, 0.04, 0, 40).to(dev)
            opt = pretrain.make_optimizer(student, clip_grad=3.0)
            images, masks, metrics = make_batch(B, seed=seed, device=dev)
            captured = {}
            orig_s, orig_t = student.forward, teacher.forward
            student.forward = lambda *a, **k: captured.setdefault("s", orig_s(*a, **k))
            teacher.forward = lambda *a, **k: captured.setdefault("t", orig_t(*a, **k))
            loss = pretrain.training_iteration(student, teacher, dino_loss, opt, images, masks, metrics, 1, 2e-4, 0.05, 0.99)
            torch.cuda.synchronize()
            so, to = O.build_pair(spec, seed=3)
            rec = O.train_iteration(so, to, torch.zeros(1, 512), O.AdamWState(), make_batch(B, seed=seed), 1, 2e-4, 0.05, 0.99)
            got = [loss.item(), dino_loss.last_losses["mask_loss"].item(), dino_loss.last_losses["Dino_loss"].item()]
            want = [rec["loss"], rec["mask_loss"], rec["dino_loss"]]
            sl = captured["s"]["instances_view"].detach().float().cpu()
            tl = captured["t"]["instances_view"].detach().float().cpu()
            osl, otl = rec["s_out"]["instances_view"].detach(), rec["t_out"]["instances_view"].detach()
            rows.append({"B": B, "seed": seed, "rows_M": int(sl.shape[0] // 2),
                         "d_loss": got[0] - want[0], "d_mask": got[1] - want[1], "d_dino": got[2] - want[2],
                         "student_logit_rms_err": float((sl - osl).pow(2).mean().sqrt()), "student_logit_rms": float(osl.pow(2).mean().sqrt()),
                         "teacher_logit_rms_err": float((tl - otl).pow(2).mean().sqrt())})
            print(json.dumps(rows[-1]), flush=True)
    by_b = {}
    for r in rows:
        by_b.setdefault(r["B"], []).append(r)
    summary = {B: {"mean_rows_M": float(np.mean([r["rows_M"] for r in v])), "rms_d_dino": float(np.sqrt(np.mean([r["d_dino"] ** 2 for r in v]))),
                   "max_abs_d_dino": float(max(abs(r["d_dino"]) for r in v)), "max_abs_d_loss": float(max(abs(r["d_loss"]) for r in v)),
                   "mean_student_logit_rms_err": float(np.mean([r["student_logit_rms_err"] for r in v]))} for B, v in by_b.items()}
    print(json.dumps({"summary_by_batch": summary}))


if __name__ == "__main__":
    main()
