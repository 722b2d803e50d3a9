import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from ccd_amd import pretrain
from ccd_amd.loss.Dino_loss import DINOLoss
from ccd_amd.synthetic import make_batch
from oracle import ccd_oracle as O
g = np.load(os.path.join(ROOT,'tests/golden/small_step.npz'))
dev = torch.device('cuda')
torch.manual_seed(0); np.random.seed(0)
student, teacher = pretrain.build_networks(arch="vit_small", out_dim=65536, drop_path_rate=0.0, norm_last_layer=False, device=dev)
dino_loss = DINOLoss(65536, 2, 0.04, 0.04, 0, 40).to(dev)
opt = pretrain.make_optimizer(student, clip_grad=3.0)
os_, ot = O.build_pair(O.Spec(norm_last_layer=False, **O.ARCH["vit_small"]), seed=0)
oc, oo = torch.zeros(1,65536), O.AdamWState()
for step in range(2):
    p=f"s{step}/"; epoch, lr, wd, mom, clip, freeze, seed = g[p+"hyper"]
    im, ma, me = make_batch(8, seed=int(seed), device=dev)
    loss = pretrain.training_iteration(student, teacher, dino_loss, opt, im, ma, me, int(epoch), lr, wd, mom, freeze_last_layer=int(freeze))
    rec = O.train_iteration(os_, ot, oc, oo, make_batch(8, seed=int(seed)), int(epoch), lr, wd, mom, freeze_last_layer=int(freeze), exact_zero_rows=True)
    oc = rec["center"]
    print("step", step, loss.item(), rec["loss"])
    for n in ["head.mlp.0.bias","head.mlp.2.bias","head.mlp.4.bias","head.last_layer.weight_g","head.last_layer.weight_v","head.mlp.4.weight"]:
        a = student.arena.g(n).float().cpu().flatten(); w = rec["grads_raw"][n].flatten()
        cos = (a@w/(a.norm()*w.norm()+1e-30)).item()
        print(f"  grad {n:28s} |hip| {a.norm():.4e} |ora| {w.norm():.4e} |diff| {(a-w).norm():.4e} cos {cos:.4f}")
    for n in ["head.mlp.0.bias","head.mlp.4.bias"]:
        a = student.arena.w(n).float().cpu().flatten(); w = os_.P[n].detach().flatten()
        print(f"  param {n:27s} |hip| {a.norm():.4e} |ora| {w.norm():.4e} |diff| {(a-w).norm():.4e} signs agree {(torch.sign(a)==torch.sign(w)).float().mean():.3f}")
