"""Elementwise gradient errors of the toy configuration against the reference fixture (what check_tiny_step bounds at 0.15)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import model_checks as mc
from ccd_amd import pretrain
from ccd_amd.loss.Dino_loss import DINOLoss
dev = torch.device("cuda:0")
for batch in (8, 2):
    g = np.load(os.path.join(mc.GOLD, "tiny_step.npz" if batch == 2 else f"tiny{batch}_step.npz"))
    student, teacher = mc.tiny_networks(dev)
    dino_loss = DINOLoss(512, 2, 0.04, 0.04, 0, 40).to(dev)
    opt = pretrain.make_optimizer(student, clip_grad=float(g["hyper"][4]))
    images, masks, metrics = mc.make_batch(batch, seed=11, device=dev)
    epoch, lr, wd, mom, clip, freeze = g["hyper"]
    loss = pretrain.training_iteration(student, teacher, dino_loss, opt, images, masks, metrics, int(epoch), lr, wd, mom, freeze_last_layer=int(freeze))
    print("batch", batch, "loss", loss.item(), "ref", g["losses"][0])
    for k in g.files:
        if k.startswith("grad/"):
            want = g[k]; got = student.arena.g(k[5:]).float().cpu().numpy().reshape(want.shape)
            d = np.abs(got - want)
            print(f"  {k:50s} max|want| {np.abs(want).max():.3e}  max err/max {d.max() / (np.abs(want).max() + 1e-12):.4f}  rms err/rms {np.sqrt((d**2).mean()) / (np.sqrt((want**2).mean()) + 1e-12):.4f}")
