#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference, read-only) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  The reference is imported
with the stub third-party modules of tools/oracle_stubs/ and its pretraining iteration (train.py:221-272)
is re-enacted by this harness; nothing from the reference is copied - only its numerical outputs for our
seeded synthetic inputs (ccd_amd/synthetic.py) are recorded.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--only ccl|tiny|small|sched]

Fixtures (all small):
  sched.npz        cosine_iter_scheduler / teacher-temp schedule arrays           (modules/utils.py:200-210)
  ccl_cases.npz    adversarial masks -> label_cluster outputs as uint8 id maps     (utils/DBSCAN.py:61-103)
  tiny_step.npz    3-block E=192 model, B=2: full tensors of every stage + grads (tiny8_step.npz: the same at B=8)
  arch_step.npz    BASELINE config #4's two architectures (vit_base = 512 / 8 heads, the 768 / 12 shape): one iteration each, B = 4
  small_step.npz   CCD_pretrain_ViT_small hyper-parameters, B=8, 2 iterations: losses, index maps,
                   sampled logits, per-parameter grad norms / post-step checksums
  small3_step.npz  the same model with perturbed head biases: 3 consecutive iterations + one at epoch 30 (predicted masks)
  finetune_step.npz  DINO_Finetune (vit_tiny/2 layers, vit_small/6 layers): 2 AdamW iterations + greedy decoding
"""
import argparse
import os
import sys
from functools import partial

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
# REPO is NOT put on sys.path: its `Dino/` alias package (a regular package) would shadow the reference's namespace package
sys.path.insert(0, os.path.join(HERE, "oracle_stubs"))
sys.path.insert(0, "/root/reference")
sys.path = [p for p in sys.path if os.path.realpath(p or os.getcwd()) != REPO]

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

import importlib.util
_spec = importlib.util.spec_from_file_location("ccd_synthetic", os.path.join(REPO, "ccd_amd", "synthetic.py"))
_syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_syn)
make_batch = _syn.make_batch              # our generators, not reference code
make_text_like_batch = _syn.make_text_like_batch
import Dino as _ref_dino
assert all(os.path.realpath(p).startswith("/root/reference") for p in _ref_dino.__path__), _ref_dino

GOLD = os.path.join(REPO, "tests", "golden")


def planes_to_idmap(planes):
    """[..., 26, H, W] 0/1 planes (disjoint) -> uint8 id map, 255 = background."""
    planes = np.asarray(planes)
    assert planes.shape[-3] == 26
    cnt = (planes > 0).sum(axis=-3)
    assert cnt.max() <= 1, "planes overlap: id-map encoding would be lossy"
    ids = np.argmax(planes > 0, axis=-3).astype(np.uint8)
    ids[cnt == 0] = 255
    return ids


def stat(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), t.pow(2).sum().sqrt().item()])


def state_stats(named):
    names, rows = [], []
    for n, t in named:
        names.append(n)
        rows.append(stat(t))
    return np.array(names), np.stack(rows)


def ensure_pg():
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)


# --------------------------------------------------------------------------------------------------
def gen_sched():
    from Dino.modules import utils as rutils
    from Dino.loss.Dino_loss import DINOLoss
    out = {}
    out["lr"] = rutils.cosine_iter_scheduler(0.0005 * 8 / 256.0, 1e-6, 50, warmup_iters=10)
    out["wd"] = rutils.cosine_iter_scheduler(0.04, 0.4, 50)
    out["mom"] = rutils.cosine_iter_scheduler(0.9995, 1, 50)
    out["lr_nowarm"] = rutils.cosine_iter_scheduler(1e-3, 1e-5, 17)
    out["teacher_temp_0_40"] = DINOLoss(16, 2, 0.04, 0.04, 0, 40).teacher_temp_schedule
    out["teacher_temp_5_12"] = DINOLoss(16, 2, 0.02, 0.07, 5, 12).teacher_temp_schedule
    np.savez_compressed(os.path.join(GOLD, "sched.npz"), **out)
    print("sched.npz written")


# --------------------------------------------------------------------------------------------------
def ccl_cases():
    H, W = 32, 128
    rs = np.random.RandomState(7)
    cases = {}
    cases["empty"] = np.zeros((H, W), np.float32)
    m = np.zeros((H, W), np.float32); m[:] = 1
    cases["full"] = m
    # areas 29 / 30 / 31 side by side
    m = np.zeros((H, W), np.float32)
    m[2:7, 2:8] = 1; m[6, 7] = 0            # 30-1 = 29
    m[2:7, 20:26] = 1                        # 30
    m[2:7, 40:46] = 1; m[7, 40] = 1          # 31
    cases["area_29_30_31"] = m
    # diagonal touching squares merge under 8-connectivity
    m = np.zeros((H, W), np.float32)
    m[4:10, 10:16] = 1; m[10:16, 16:22] = 1
    m[4:10, 60:66] = 1; m[11:17, 66:72] = 1  # gap of one row: stays separate
    cases["diagonal"] = m
    # 40 one-pixel stripes, 32 tall -> more than 26 qualifying components
    m = np.zeros((H, W), np.float32)
    for k in range(40):
        m[:, 3 * k] = 1
    cases["stripes40"] = m
    # stripes whose raster-first-pixel order differs from left-to-right order (cap keeps first 26 labels)
    m = np.zeros((H, W), np.float32)
    for k in range(32):
        top = (k * 7) % 2
        m[top:32, 4 * k] = 1
    cases["stripes32_stagger"] = m
    # thin stroke invisible to the centre-2x2 down-sample, plus a normal char
    m = np.zeros((H, W), np.float32)
    m[0:32, 4] = 1      # column 4: x%4==0 -> never in centre columns (1,2)
    m[8:24, 40:50] = 1
    cases["thin_stroke"] = m
    # U shape / nested ring
    m = np.zeros((H, W), np.float32)
    m[4:28, 10:40] = 1; m[8:24, 14:36] = 0; m[12:20, 20:30] = 1
    cases["ring_nested"] = m
    # spiral-ish snake to stress label propagation
    m = np.zeros((H, W), np.float32)
    for r in range(0, 32, 4):
        m[r, 2:126] = 1
        if (r // 4) % 2 == 0:
            m[r:r + 4, 125] = 1
        else:
            m[r:r + 4, 2] = 1
    cases["snake"] = m
    # random blobs at several densities
    for i, p in enumerate([0.3, 0.45, 0.55, 0.7]):
        cases[f"random_p{int(p * 100)}"] = (rs.uniform(size=(H, W)) < p).astype(np.float32)
    # checkerboard: all diagonal-connected -> one component
    yy, xx = np.mgrid[0:H, 0:W]
    cases["checker"] = ((yy + xx) % 2 == 0).astype(np.float32)
    # blocky random (characters-like)
    m = np.zeros((H, W), np.float32)
    for _ in range(30):
        y0, x0 = rs.randint(0, 26), rs.randint(0, 120)
        m[y0:y0 + rs.randint(3, 9), x0:x0 + rs.randint(3, 9)] = 1
    cases["blocks30"] = m
    return cases


def gen_ccl():
    from Dino.utils.DBSCAN import label_cluster
    lab = label_cluster()
    cases = ccl_cases()
    # mean-x ties make the reference's argsort order host dependent: record a flag
    names, masks, idmaps, has_tie = [], [], [], []
    from scipy import ndimage
    for name, m in cases.items():
        planes = lab(m)
        names.append(name); masks.append(m.astype(np.uint8)); idmaps.append(planes_to_idmap(planes))
        cl = ndimage.label(m != 0, structure=np.ones((3, 3)))[0]
        locs = []
        for c in range(1, cl.max() + 1):
            sub = cl == c
            if sub.sum() >= 30:
                locs.append(np.where(sub)[1].mean())
                if len(locs) >= 26:
                    break
        has_tie.append(len(set(locs)) != len(locs))
    np.savez_compressed(os.path.join(GOLD, "ccl_cases.npz"), names=np.array(names), masks=np.stack(masks),
                        idmaps=np.stack(idmaps), has_tie=np.array(has_tie))
    print("ccl_cases.npz written:", dict(zip(names, has_tie)))


# --------------------------------------------------------------------------------------------------
def build_reference_pair(arch_kwargs, head_kwargs, seg_channel, seed, drop_path_rate, tiny):
    """Mirrors the construction ORDER of train.py:63-114 (student vit, teacher vit, SegHead, DINOHeads)."""
    from Dino.modules import vision_transformer as vits
    from Dino.modules.segmentor import SegHead
    from Dino.model.dino_vision import ABIDINOModel
    torch.manual_seed(seed)
    np.random.seed(seed)
    if tiny:
        mk = lambda **kw: vits.VisionTransformer(img_size=[32, 128], qkv_bias=True,
                                                 norm_layer=partial(nn.LayerNorm, eps=1e-6), **arch_kwargs, **kw)
    else:
        mk = lambda **kw: vits.__dict__[arch_kwargs["arch"]](patch_size=4, **kw)
    student_b = mk(drop_path_rate=drop_path_rate)
    teacher_b = mk()
    embed_dim = student_b.embed_dim
    student = ABIDINOModel(student_b, SegHead(in_channels=seg_channel, mla_channels=128, mlahead_channels=64,
                                              num_classes=2),
                           vits.DINOHead(embed_dim, norm_last_layer=False, **head_kwargs))
    teacher = ABIDINOModel(teacher_b, None, vits.DINOHead(embed_dim, **head_kwargs))
    teacher.backbone.load_state_dict(student.backbone.state_dict())
    teacher.head.load_state_dict(student.head.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    return student, teacher


def reference_iteration(student, teacher, dino_loss, optimizer, batch, epoch, lr, wd, mom, clip, freeze_last_layer,
                        record):
    """train.py:221-272 re-enacted on CPU tensors."""
    from Dino.modules import utils as rutils
    images, masks, metrics = batch
    for i, g in enumerate(optimizer.param_groups):
        g["lr"] = lr
        if i == 0:
            g["weight_decay"] = wd
    s_out = student(images, metrics, masks, epoch, clusters=None)
    t_out = teacher(images, metrics, None, None, clusters=s_out["zero"], index=s_out["index"])
    grid = F.affine_grid(metrics[:, :2, :], size=(masks.shape[0], 1, masks.shape[1], masks.shape[2]))
    masks_image = (F.grid_sample(masks.unsqueeze(1), grid) > 0.1).float().squeeze()
    s_out["gt"] = [masks, masks_image]
    center_before = dino_loss.center.clone()
    loss = dino_loss(s_out, t_out, epoch)
    optimizer.zero_grad()
    loss.backward()
    record["loss"] = loss.item()
    record["mask_loss"] = dino_loss.last_losses["mask_loss"].item()
    record["dino_loss"] = dino_loss.last_losses["Dino_loss"].item()
    record["s_out"], record["t_out"] = s_out, t_out
    record["masks_image"] = masks_image
    record["center_before"] = center_before
    record["grads_raw"] = {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}
    rutils.clip_gradients(student, clip)
    record["grads_clipped"] = {n: p.grad.detach().clone() for n, p in student.named_parameters()
                               if p.grad is not None}
    rutils.cancel_gradients_last_layer(epoch, student, freeze_last_layer)
    optimizer.step()
    with torch.no_grad():
        for pq, pk in zip(student.backbone.parameters(), teacher.backbone.parameters()):
            pk.data.mul_(mom).add_((1 - mom) * pq.detach().data)
        for pq, pk in zip(student.head.parameters(), teacher.head.parameters()):
            pk.data.mul_(mom).add_((1 - mom) * pq.detach().data)
    return record


def gen_tiny():
    # B = 2 (tiny_step.npz: what the CPU SIMT executor can afford, three times per CPU test run) and, round 4, B = 8 (tiny8_step.npz: the
    # GPU parity gate at the north-star tolerance - at B = 2 the loss averages ~14 selected rows and their bf16 logit noise does not
    # average out: profiles/r04_parity_tiny_budget.json)
    for B, name in ((2, "tiny_step.npz"), (8, "tiny8_step.npz")):
        gen_tiny_batch(B, name)


def gen_tiny_batch(B, name):
    ensure_pg()
    from Dino.modules import utils as rutils
    from Dino.loss.Dino_loss import DINOLoss
    K = 512
    arch = dict(patch_size=4, embed_dim=192, depth=3, num_heads=3, out_indices=[1, 2, 3])
    head = dict(out_dim=K, hidden_dim=256, bottleneck_dim=64)
    student, teacher = build_reference_pair(arch, head, 192, seed=3, drop_path_rate=0.0, tiny=True)
    out = {}
    out["init_names"], out["init_stats"] = state_stats(student.state_dict().items())
    dino_loss = DINOLoss(K, 2, 0.04, 0.04, 0, 40)
    optimizer = torch.optim.AdamW(rutils.get_params_groups(student))
    batch = make_batch(B, seed=11)
    rec = reference_iteration(student, teacher, dino_loss, optimizer, batch, epoch=1, lr=2e-4, wd=0.05, mom=0.99,
                              clip=3.0, freeze_last_layer=1, record={})
    # stage tensors (full, the model is small)
    images, masks, metrics = batch
    s_out, t_out = rec["s_out"], rec["t_out"]
    out["masks"] = masks.numpy().astype(np.uint8)
    out["metrics"] = metrics.numpy()
    out["zero_idmap"] = planes_to_idmap(s_out["zero"].numpy())
    out["new_index"] = s_out["index"].numpy()
    out["masks_image"] = rec["masks_image"].numpy().astype(np.uint8)
    out["seg_logits"] = s_out["mask"].detach().numpy()
    out["student_logits"] = s_out["instances_view"].detach().numpy()
    out["teacher_logits"] = t_out["instances_view"].detach().numpy()
    out["teacher_feature"] = t_out["feature"].detach().numpy()[:, ::4]  # every 4th channel, [2B,48,8,32]
    out["losses"] = np.array([rec["loss"], rec["mask_loss"], rec["dino_loss"]])
    out["center_after"] = dino_loss.center.numpy()
    gn, gs = state_stats(rec["grads_raw"].items())
    out["grad_names"], out["grad_stats"] = gn, gs
    # a few full gradients for layout checks
    for key in ["backbone.pos_embed", "backbone.patch_embed.proj.weight", "backbone.blocks.1.attn.qkv.weight",
                "backbone.blocks.2.mlp.fc2.bias", "backbone.norm_seg.1.weight", "segmentation.cls.weight",
                "segmentation.mlahead.head3.3.weight", "segmentation.unpool1.1.weight", "head.mlp.4.weight",
                "head.last_layer.weight_g"]:
        out["grad/" + key] = rec["grads_raw"][key].numpy()
    out["post_names"], out["post_stats"] = state_stats(student.state_dict().items())
    out["teacher_post_names"], out["teacher_post_stats"] = state_stats(teacher.state_dict().items())
    out["hyper"] = np.array([1, 2e-4, 0.05, 0.99, 3.0, 1])  # epoch lr wd mom clip freeze
    np.savez_compressed(os.path.join(GOLD, name), **out)
    print(name, "written; losses", out["losses"], "M", int(out["new_index"].sum()))


def gen_small():
    ensure_pg()
    from Dino.modules import utils as rutils
    from Dino.loss.Dino_loss import DINOLoss
    B, K = 8, 65536
    student, teacher = build_reference_pair(dict(arch="vit_small"), dict(out_dim=K), 384, seed=0,
                                            drop_path_rate=0.0, tiny=False)
    out = {}
    out["init_names"], out["init_stats"] = state_stats(student.state_dict().items())
    # A segmentation head WITH margins.  Straight from its random init the head's two logits differ by ~0 at every pixel, so the
    # thresholded prediction is decided by rounding noise and no reduced-precision implementation can reproduce it.  The state
    # the fixture starts from is therefore a head whose last layer (segmentation.cls, 2 x 128 x 3 x 3 + 2 numbers - small
    # enough to record) has been fitted to the text masks of the epoch-30 batch on the head's own features: the prediction
    # then has real foreground / background margins, as a head that has trained for 30 epochs would.
    out["cls_weight"], out["cls_bias"] = fit_seg_classifier(student, make_text_like_batch(B, seed=13))
    dino_loss = DINOLoss(K, 2, 0.04, 0.04, 0, 40)
    optimizer = torch.optim.AdamW(rutils.get_params_groups(student))
    lr_s = rutils.cosine_iter_scheduler(0.0005 * B / 256.0, 1e-6, 50, warmup_iters=10)
    wd_s = rutils.cosine_iter_scheduler(0.04, 0.4, 50)
    mom_s = rutils.cosine_iter_scheduler(0.9995, 1, 50)
    rows = np.array([0, 5, 17, 40, 47, 63, 80, 95])
    cols = np.arange(0, K, 1024)
    for step, (it, epoch, seed) in enumerate([(5, 0, 0), (6, 1, 1)]):
        batch = make_batch(B, seed=seed)
        rec = reference_iteration(student, teacher, dino_loss, optimizer, batch, epoch=epoch, lr=lr_s[it],
                                  wd=wd_s[it], mom=mom_s[it], clip=3.0, freeze_last_layer=1, record={})
        p = f"s{step}/"
        s_out, t_out = rec["s_out"], rec["t_out"]
        out[p + "hyper"] = np.array([epoch, lr_s[it], wd_s[it], mom_s[it], 3.0, 1, seed])
        out[p + "masks"] = batch[1].numpy().astype(np.uint8)
        out[p + "metrics"] = batch[2].numpy()
        out[p + "image_stat"] = stat(batch[0])
        out[p + "zero_idmap"] = planes_to_idmap(s_out["zero"].numpy())
        out[p + "new_index"] = s_out["index"].numpy()
        out[p + "masks_image"] = rec["masks_image"].numpy().astype(np.uint8)
        out[p + "losses"] = np.array([rec["loss"], rec["mask_loss"], rec["dino_loss"]])
        sl, tl = s_out["instances_view"].detach(), t_out["instances_view"].detach()
        r = rows[rows < sl.shape[0]]
        out[p + "rows"], out[p + "cols"] = r, cols
        out[p + "student_logits_sample"] = sl[r][:, cols].numpy()
        out[p + "teacher_logits_sample"] = tl[r][:, cols].numpy()
        out[p + "student_logits_stat"] = stat(sl)
        out[p + "teacher_logits_stat"] = stat(tl)
        out[p + "seg_logits_stat"] = stat(s_out["mask"])
        out[p + "seg_logits_sample"] = s_out["mask"].detach()[:, :, ::8, ::16].numpy()
        out[p + "teacher_feature_stat"] = stat(t_out["feature"])
        out[p + "center_stat"] = stat(dino_loss.center)
        out[p + "center_sample"] = dino_loss.center[0, cols].numpy()
        out[p + "grad_names"], out[p + "grad_stats"] = state_stats(rec["grads_raw"].items())
        _, out[p + "grad_clipped_stats"] = state_stats(rec["grads_clipped"].items())
        out[p + "post_names"], out[p + "post_stats"] = state_stats(student.state_dict().items())
        out[p + "teacher_post_names"], out[p + "teacher_post_stats"] = state_stats(teacher.state_dict().items())
        print(f"step {step}: losses {out[p + 'losses']}  M={int(s_out['index'].sum())}")
    # predicted-mask branch (epoch >= 30, dino_vision.py:64-70): record the thresholded prediction + its maps
    batch = make_batch(B, seed=2)
    with torch.no_grad():
        s_out = student(batch[0], batch[2], batch[1], 30, clusters=None)
        pred = (F.softmax(s_out["mask"], dim=1)[:, 1] > 0.5).int()[:B]
    out["pred/mask"] = pred.numpy().astype(np.uint8)
    out["pred/metrics"] = batch[2].numpy()
    out["pred/zero_idmap"] = planes_to_idmap(s_out["zero"].numpy())
    out["pred/new_index"] = s_out["index"].numpy()
    np.savez_compressed(os.path.join(GOLD, "small_step.npz"), **out)
    print("small_step.npz written")


def gen_small_noise():
    """How far does the REFERENCE's own iteration-1 loss of small_step.npz move when only its summation order changes?  The same two
    iterations as the committed small_step.npz (its starting state: iteration 0 reproduces the recorded losses) under 1 / 2 / 4 / 8 intra-op threads -
    oneDNN / ATen split their reductions by thread count.  At iteration 0 the head biases are exactly zero, the all-zero pooled row
    of every image reaches F.normalize as an exact zero vector, and its backward multiplies fp32 rounding residue by 1 / eps: the
    first update of three head-bias tensors is amplified noise, and every later loss inherits it.  The spread
    recorded here is the band tests/model_checks.py::check_small_steps allows between the HIP path and the recorded reference at
    iteration 1 (it was a chosen 2e-2 before).  Thread counts turned out NOT to move it (the residue comes from row-wise softmax
    arithmetic, which is not split across threads); permuting the K output units of the last layer - the same permutation in both
    networks, an exact symmetry of the loss - does: it changes the order of every sum over k."""
    import json
    ensure_pg()
    from Dino.modules import utils as rutils
    from Dino.loss.Dino_loss import DINOLoss
    B, K = 8, 65536
    g = np.load(os.path.join(GOLD, "small_step.npz"))
    runs = []
    # (threads, permutation seed): a permutation of the K output units of the last layer - the same one for student and teacher - is
    # an exact symmetry of the loss (softmax, centre and cross entropy are sums over k); it only changes the ORDER of those sums
    # ... and, last, the loss's log_softmax evaluated in float64 (this harness swaps the function the reference's loss module calls;
    # nothing of the reference changes): the residue is exp(log_softmax(0)) - softmax(0) = -2.3e-12 per logit of an exactly-uniform
    # row - a property of THIS platform's fp32 expf / logf, the same for every k, hence independent of any summation order - and in
    # float64 it is ~1e-17: the amplified head-bias gradients vanish and iteration 1 lands where an exact implementation lands
    import torch.nn.functional as tF
    fp32_log_softmax = tF.log_softmax
    for threads, perm_seed, ls64 in ((1, None, False), (2, None, False), (4, None, False), (8, None, False), (8, 1, False), (8, 2, False),
                                     (8, 3, False), (8, None, True)):
        tF.log_softmax = (lambda x, dim=None, **kw: fp32_log_softmax(x.double(), dim=dim).to(x.dtype)) if ls64 else fp32_log_softmax
        torch.set_num_threads(threads)
        student, teacher = build_reference_pair(dict(arch="vit_small"), dict(out_dim=K), 384, seed=0, drop_path_rate=0.0, tiny=False)
        if perm_seed is not None:
            perm = torch.randperm(K, generator=torch.Generator().manual_seed(perm_seed))
            with torch.no_grad():
                for net in (student, teacher):
                    for name in ("weight_v", "weight_g"):
                        t = getattr(net.head.last_layer, name)
                        t.copy_(t[perm].clone())
        if "cls_weight" in g.files:        # (the committed small_step.npz starts from the plain seed-0 state: iteration 0 below must
            with torch.no_grad():          # reproduce its recorded losses, which is checked)
                student.segmentation.cls.weight.copy_(torch.from_numpy(g["cls_weight"]))
                student.segmentation.cls.bias.copy_(torch.from_numpy(g["cls_bias"]))
        dino_loss = DINOLoss(K, 2, 0.04, 0.04, 0, 40)
        optimizer = torch.optim.AdamW(rutils.get_params_groups(student))
        lr_s = rutils.cosine_iter_scheduler(0.0005 * B / 256.0, 1e-6, 50, warmup_iters=10)
        wd_s = rutils.cosine_iter_scheduler(0.04, 0.4, 50)
        mom_s = rutils.cosine_iter_scheduler(0.9995, 1, 50)
        row = {"threads": threads, "output_permutation_seed": perm_seed, "loss_log_softmax_in_float64": ls64}
        for step, (it, epoch, seed) in enumerate([(5, 0, 0), (6, 1, 1)]):
            rec = reference_iteration(student, teacher, dino_loss, optimizer, make_batch(B, seed=seed), epoch=epoch, lr=lr_s[it],
                                      wd=wd_s[it], mom=mom_s[it], clip=3.0, freeze_last_layer=1, record={})
            row[f"s{step}"] = [float(rec["loss"]), float(rec["mask_loss"]), float(rec["dino_loss"])]
            head_bias = [n for n in rec["grads_raw"] if n.startswith("head.mlp") and n.endswith("bias")]
            row[f"s{step}_head_bias_grad_l2"] = {n: float(rec["grads_raw"][n].float().norm()) for n in head_bias}
        print(row)
        tF.log_softmax = fp32_log_softmax
        assert np.abs(np.array(row["s0"]) - g["s0/losses"]).max() < 1e-4, ("not the fixture's starting state", row["s0"], g["s0/losses"])
        runs.append(row)
    torch.set_num_threads(8)
    fix = [float(x) for x in g["s1/losses"]]
    exact = [r for r in runs if r["loss_log_softmax_in_float64"]]
    runs32 = [r for r in runs if not r["loss_log_softmax_in_float64"]]
    s1 = np.array([r["s1"] for r in runs32])
    out = {"what": "reference losses [total, mask, dino] of small_step.npz's two iterations under 1/2/4/8 intra-op threads and under "
                   "three permutations of the last layer's 65536 output units (an exact symmetry of the loss: only summation orders change)",
           "runs": runs, "fixture_s1_losses": fix,
           "s0_spread": (np.array([r["s0"] for r in runs32]).max(0) - np.array([r["s0"] for r in runs32]).min(0)).tolist(),
           "s1_losses_with_float64_log_softmax": exact[0]["s1"] if exact else None,
           "s1_shift_fp32_vs_float64_log_softmax": (np.array(exact[0]["s1"]) - s1[0]).tolist() if exact else None,
           "s1_spread": (s1.max(0) - s1.min(0)).tolist(),
           "s1_max_abs_from_fixture": np.abs(s1 - np.array(fix)).max(0).tolist()}
    with open(os.path.join(GOLD, "small_step_ref_noise.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("small_step_ref_noise.json written:", out["s1_spread"], out["s1_max_abs_from_fixture"])


HEAD_BIAS_PERTURB = dict(seed=1234, scale=0.02)


def perturb_head_biases(named_parameters):
    """In place: every bias of the DINO head += scale * N(0,1) from a seeded generator, in parameter order.  With non-zero
    head biases no pooled row reaches F.normalize as an exact zero vector, so the reference's head-bias gradients are not
    rounding residue amplified by 1/eps and multi-iteration parity can be asserted tensor by tensor."""
    g = torch.Generator().manual_seed(HEAD_BIAS_PERTURB["seed"])
    with torch.no_grad():
        for n, p in named_parameters:
            if n.startswith("head.") and n.endswith(".bias"):
                p.add_(HEAD_BIAS_PERTURB["scale"] * torch.randn(p.shape, generator=g))


def fit_seg_classifier(student, batch, steps=400):
    """Fit ONLY student.segmentation.cls (in place) so that softmax(cls(features))[:, 1] predicts the text masks of view 1 of
    `batch`: features = the input of `cls` in a training-mode forward (BatchNorm batch statistics - what the recorded
    iteration will see); plain logistic regression with L-BFGS in this script (not reference code).  Buffers touched by
    the extra forward (BatchNorm running statistics) are restored.  Returns the fitted (weight, bias) as numpy arrays."""
    images, masks, metrics = batch
    B = images.shape[0]
    buffers = {k: v.clone() for k, v in student.state_dict().items()}
    feats = {}
    hook = student.segmentation.cls.register_forward_hook(lambda m, i, o: feats.__setitem__("x", i[0].detach()))
    student.train()
    with torch.no_grad():
        student(images, metrics, masks, 0, clusters=None)
    hook.remove()
    student.load_state_dict(buffers)
    x = feats["x"][:B]                                           # the branch thresholds the first B images (view 1)
    target = masks.long()
    cls = student.segmentation.cls
    w = cls.weight.detach().clone().requires_grad_(True)
    b = cls.bias.detach().clone().requires_grad_(True)
    opt = torch.optim.LBFGS([w, b], lr=1.0, max_iter=steps, history_size=20, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = F.cross_entropy(F.conv2d(x, w, b, padding=1), target) + 1e-4 * (w * w).sum()
        loss.backward()
        return loss

    opt.step(closure)
    with torch.no_grad():
        cls.weight.copy_(w)
        cls.bias.copy_(b)
        logit = F.conv2d(x, w, b, padding=1)
        margin = (logit[:, 1] - logit[:, 0])
        acc = ((margin > 0) == (target > 0)).float().mean().item()
        q = torch.quantile(margin.abs().flatten(), torch.tensor([0.001, 0.01, 0.1, 0.5]))
        print(f"fit_seg_classifier: pixel accuracy {acc:.4f}, |margin| quantiles 0.1% / 1% / 10% / 50%: {q.tolist()}, "
              f"pixels with |margin| < 0.05: {int((margin.abs() < 0.05).sum())} of {margin.numel()}")
    return cls.weight.detach().numpy().copy(), cls.bias.detach().numpy().copy()


def gen_small3():
    """CCD_pretrain_ViT_small, B=8, head biases perturbed before the first step: one iteration at epoch 30 (predicted-mask
    branch, dino_vision.py:64-70, on images that carry the characters) followed by THREE consecutive iterations on the
    dataset-mask branch, all on the same model and optimizer state -> small3_step.npz."""
    ensure_pg()
    from Dino.modules import utils as rutils
    from Dino.loss.Dino_loss import DINOLoss
    B, K = 8, 65536
    student, teacher = build_reference_pair(dict(arch="vit_small"), dict(out_dim=K), 384, seed=0,
                                            drop_path_rate=0.0, tiny=False)
    perturb_head_biases(student.named_parameters())
    teacher.head.load_state_dict(student.head.state_dict())
    out = {"perturb": np.array([HEAD_BIAS_PERTURB["seed"], HEAD_BIAS_PERTURB["scale"]])}
    out["init_names"], out["init_stats"] = state_stats(student.state_dict().items())
    # A segmentation head WITH margins.  Straight from its random init the head's two logits differ by ~0 at every pixel, so the
    # thresholded prediction is decided by rounding noise and no reduced-precision implementation can reproduce it.  The state
    # the fixture starts from is therefore a head whose last layer (segmentation.cls, 2 x 128 x 3 x 3 + 2 numbers - small
    # enough to record) has been fitted to the text masks of the epoch-30 batch on the head's own features: the prediction
    # then has real foreground / background margins, as a head that has trained for 30 epochs would.
    out["cls_weight"], out["cls_bias"] = fit_seg_classifier(student, make_text_like_batch(B, seed=13))
    dino_loss = DINOLoss(K, 2, 0.04, 0.04, 0, 40)
    optimizer = torch.optim.AdamW(rutils.get_params_groups(student))
    lr_s = rutils.cosine_iter_scheduler(0.0005 * B / 256.0, 1e-6, 50, warmup_iters=10)
    wd_s = rutils.cosine_iter_scheduler(0.04, 0.4, 50)
    mom_s = rutils.cosine_iter_scheduler(0.9995, 1, 50)
    rows = np.array([0, 5, 17, 40, 47, 63, 80, 95])
    cols = np.arange(0, K, 1024)
    # the predicted-mask step comes FIRST: a few AdamW steps on the mask loss push an untrained segmentation head to
    # "background everywhere" (75 % of the pixels are background), after which the branch has no component left to label
    for step, (it, epoch, seed) in enumerate([(5, 30, 13), (6, 0, 10), (7, 0, 11), (8, 1, 12)]):
        # the predicted-mask step runs on images that carry the characters (pure noise gives no 30-pixel component)
        batch = make_text_like_batch(B, seed=seed) if epoch >= 30 else make_batch(B, seed=seed)
        rec = reference_iteration(student, teacher, dino_loss, optimizer, batch, epoch=epoch, lr=lr_s[it],
                                  wd=wd_s[it], mom=mom_s[it], clip=3.0, freeze_last_layer=1, record={})
        p = f"s{step}/"
        s_out, t_out = rec["s_out"], rec["t_out"]
        out[p + "hyper"] = np.array([epoch, lr_s[it], wd_s[it], mom_s[it], 3.0, 1, seed])
        out[p + "masks"] = batch[1].numpy().astype(np.uint8)
        out[p + "metrics"] = batch[2].numpy()
        out[p + "zero_idmap"] = planes_to_idmap(s_out["zero"].numpy())
        out[p + "new_index"] = s_out["index"].numpy()
        out[p + "losses"] = np.array([rec["loss"], rec["mask_loss"], rec["dino_loss"]])
        sl, tl = s_out["instances_view"].detach(), t_out["instances_view"].detach()
        r = rows[rows < sl.shape[0]]
        out[p + "rows"], out[p + "cols"] = r, cols
        out[p + "student_logits_sample"] = sl[r][:, cols].numpy()
        out[p + "teacher_logits_sample"] = tl[r][:, cols].numpy()
        out[p + "center_stat"] = stat(dino_loss.center)
        out[p + "grad_names"], out[p + "grad_stats"] = state_stats(rec["grads_raw"].items())
        out[p + "post_names"], out[p + "post_stats"] = state_stats(student.state_dict().items())
        out[p + "teacher_post_names"], out[p + "teacher_post_stats"] = state_stats(teacher.state_dict().items())
        if epoch >= 30:     # the branch thresholds the student's own segmentation: keep the logits of view 1 in full
            seg = s_out["mask"].detach()[:B]
            out[p + "seg_logits_view1"] = seg.numpy()
            out[p + "pred_mask"] = (F.softmax(seg, dim=1)[:, 1] > 0.5).numpy().astype(np.uint8)
            out[p + "pred_margin_min"] = np.array([(seg[:, 1] - seg[:, 0]).abs().min().item()])
        print(f"small3 step {step} (epoch {epoch}): losses {out[p + 'losses']}  M={int(s_out['index'].sum())}"
              f"  planes/img={[int((np.unique(m) != 255).sum()) for m in out[p + 'zero_idmap'][:B]]}")
    np.savez_compressed(os.path.join(GOLD, "small3_step.npz"), **out)
    print("small3_step.npz written")


def gen_arch():
    """One reference iteration of BASELINE config #4's architectures: vit_base (the reference's factory,
    vision_transformer.py:287-291: 512 / 8 heads) and the 768 / 12 shape the config's text names (the VisionTransformer
    constructor's defaults), B = 4, out_dim 4096, hyper-parameters of tests/model_checks.py::check_pretrain_arch_vs_oracle."""
    ensure_pg()
    from Dino.modules import utils as rutils
    from Dino.loss.Dino_loss import DINOLoss
    B, K = 4, 4096
    out = {}
    for arch in ("vit_base", "vit_base_768"):
        if arch == "vit_base_768":
            student, teacher = build_reference_pair(dict(embed_dim=768, depth=12, num_heads=12, patch_size=4), dict(out_dim=K), 768,
                                                    seed=0, drop_path_rate=0.0, tiny=True)
        else:
            student, teacher = build_reference_pair(dict(arch=arch), dict(out_dim=K), 512, seed=0, drop_path_rate=0.0, tiny=False)
        p = arch + "/"
        out[p + "init_names"], out[p + "init_stats"] = state_stats(student.state_dict().items())
        dino_loss = DINOLoss(K, 2, 0.04, 0.04, 0, 40)
        optimizer = torch.optim.AdamW(rutils.get_params_groups(student))
        batch = make_batch(B, seed=21)
        rec = reference_iteration(student, teacher, dino_loss, optimizer, batch, epoch=1, lr=1e-4, wd=0.04, mom=0.9995,
                                  clip=3.0, freeze_last_layer=1, record={})
        s_out, t_out = rec["s_out"], rec["t_out"]
        out[p + "hyper"] = np.array([1, 1e-4, 0.04, 0.9995, 3.0, 1, 21, B, K])
        out[p + "zero_idmap"] = planes_to_idmap(s_out["zero"].numpy())
        out[p + "new_index"] = s_out["index"].numpy()
        out[p + "losses"] = np.array([rec["loss"], rec["mask_loss"], rec["dino_loss"]])
        out[p + "student_logits_stat"] = stat(s_out["instances_view"].detach())
        out[p + "teacher_logits_stat"] = stat(t_out["instances_view"].detach())
        out[p + "center_after"] = dino_loss.center.numpy()
        out[p + "grad_names"], out[p + "grad_stats"] = state_stats(rec["grads_raw"].items())
        print(arch, "losses", out[p + "losses"], "M", int(out[p + "new_index"].sum()))
    np.savez_compressed(os.path.join(GOLD, "arch_step.npz"), **out)
    print("arch_step.npz written")


def gen_keys():
    """State-dict key names + shapes of student/teacher for the three shipped archs."""
    import json
    table = {}
    for arch in ["vit_tiny", "vit_small", "vit_base", "vit_base_768"]:
        from Dino.modules import vision_transformer as vits
        e = {"vit_tiny": 192, "vit_small": 384, "vit_base": 512, "vit_base_768": 768}[arch]
        if arch == "vit_base_768":
            # BASELINE config #4's shape: not a factory of the reference, but what its VisionTransformer constructor builds by
            # default (embed_dim=768, depth=12, num_heads=12, vision_transformer.py:117-120) - with qkv_bias / eps as the factories set them
            student, teacher = build_reference_pair(dict(embed_dim=768, depth=12, num_heads=12, patch_size=4), dict(out_dim=1024), e,
                                                    seed=0, drop_path_rate=0.1, tiny=True)
        else:
            student, teacher = build_reference_pair(dict(arch=arch), dict(out_dim=1024), e, seed=0, drop_path_rate=0.1,
                                                    tiny=False)
        table[arch] = {
            "student": [[k, list(v.shape), str(v.dtype)] for k, v in student.state_dict().items()],
            "teacher": [[k, list(v.shape), str(v.dtype)] for k, v in teacher.state_dict().items()],
            "student_trainable": [n for n, p in student.named_parameters() if p.requires_grad],
        }
    with open(os.path.join(GOLD, "state_keys.json"), "w") as f:
        json.dump(table, f)
    print("state_keys.json written", {k: len(v["student"]) for k, v in table.items()})


FT_WORDS = ["hello", "Wor1d!", "MI355X", "a", "text-recognition", "CCD", "~{unknown}", "0123456789abcdefghijklmnopqrstuvwxyz"]


def build_reference_finetune(arch, n_layers, seed):
    """DINO_Finetune (Dino/model/dino_vision.py:134-185) with every nn.Dropout set to p = 0 and drop_path_rate 0."""
    from Dino.model.dino_vision import DINO_Finetune

    class Cfg:
        pass
    c = Cfg()
    c.arch, c.patch_size, c.drop_path_rate = arch, 4, 0.0
    c.decoder_max_seq_len, c.decoder_n_layers, c.decoder_d_embedding, c.decoder_n_head = 25, n_layers, 512, 8
    c.decoder_d_k = c.decoder_d_v = 64
    c.decoder_d_model, c.decoder_d_inner = 512, 256
    torch.manual_seed(seed)
    model = DINO_Finetune(c)
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model.train()
    return model


def gen_finetune():
    """finetune_step.npz: two iterations of train_finetune.py:262-289 (AdamW, no clipping) + greedy decoding, for
    vit_tiny/2 decoder layers (full tensors) and vit_small/6 layers (the shipped CCD_vision_model_ARD configuration)."""
    from Dino.modules import utils as rutils
    out = {"words": np.array(FT_WORDS)}
    out["sched"] = rutils.cosine_scheduler(0.0005, 1e-6, 3, 20, warmup_epochs=1)
    for tag, arch, n_layers, B in (("tiny", "vit_tiny", 2, 4), ("small", "vit_small", 6, 8)):
        model = build_reference_finetune(arch, n_layers, seed=0)
        names, stats = state_stats(model.state_dict().items())
        out[f"{tag}/init_names"], out[f"{tag}/init_stats"] = names, stats
        targets = model.label_convertor.str2tensor(FT_WORDS[:B])
        out[f"{tag}/targets"] = targets.numpy()
        opt = torch.optim.AdamW(rutils.get_params_groups(model), lr=0.0005, betas=(0.9, 0.999), weight_decay=0.05)
        g = torch.Generator().manual_seed(1234)
        for step in range(2):
            img = torch.randn(B, 3, 32, 128, generator=g)
            lr = float(out["sched"][step + 3])
            for grp in opt.param_groups:
                grp["lr"] = lr
            losses, attn = model(img, targets, return_loss=True)
            loss = losses.mean()
            model.zero_grad()
            loss.backward()
            opt.step()
            p = f"{tag}/s{step}/"
            out[p + "image_stat"] = stat(img)
            out[p + "loss"] = np.array([loss.item(), lr])
            out[p + "attn_mean"] = attn.detach().mean(1).numpy().astype(np.float32)
            gn, gs = state_stats([(n, q.grad) for n, q in model.named_parameters() if q.grad is not None])
            out[p + "grad_names"], out[p + "grad_stats"] = gn, gs
            pn, ps = state_stats(model.named_parameters())
            out[p + "post_names"], out[p + "post_stats"] = pn, ps
        # train-mode logits + greedy decoding on a fresh batch with the updated weights
        img = torch.randn(B, 3, 32, 128, generator=g)
        with torch.no_grad():
            feat = model.extract_feat(img)
            logits, _ = model.decoder(feat, model.encoder(feat), {"padded_targets": targets}, train_mode=True)
            model.eval()
            probs = model(img, None, return_loss=False)
        out[f"{tag}/eval_image_stat"] = stat(img)
        out[f"{tag}/logits"] = logits.numpy().astype(np.float32)
        out[f"{tag}/test_probs"] = probs.numpy().astype(np.float32)
        print(tag, "losses", out[f"{tag}/s0/loss"], out[f"{tag}/s1/loss"], "decoded", probs.argmax(-1)[0, :8].tolist())
    np.savez_compressed(os.path.join(GOLD, "finetune_step.npz"), **out)
    print("finetune_step.npz written")


def gen_eval():
    """eval_acc.npz: the REAL `TextAccuracy.compute` (Dino/metric/eval_acc.py:27-64, as test.py:198-203 drives it) on the
    seeded vit_tiny recogniser: predictions of three batches, ground truths derived from them (exact, case / punctuation
    variants, edits, unrelated words), and the metric dictionary."""
    from Dino.metric.eval_acc import TextAccuracy
    model = build_reference_finetune("vit_tiny", 2, seed=0)
    model.eval()
    g = torch.Generator().manual_seed(4321)
    batches = [torch.randn(6, 3, 32, 128, generator=g) for _ in range(3)]
    with torch.no_grad():
        preds = []
        for img in batches:
            idx, _ = model.label_convertor.tensor2idx(model(img, None, return_loss=False))
            preds.append(model.label_convertor.idx2str(idx))

    def variant(k, p):
        if k % 6 == 0:
            return p                                              # exact
        if k % 6 == 1:
            return p.swapcase()                                   # case differs only
        if k % 6 == 2:
            return (p[:2] + "-" + p[2:] + "!") if p else "!"       # punctuation the normalisation removes
        if k % 6 == 3:
            return (p[:-1] + "x") if p else "x"                   # one substitution
        if k % 6 == 4:
            return "unrelated" + str(k)
        return p[1:] if len(p) > 1 else p + "q"                   # one deletion
    gts = [[variant(6 * b + i, p) for i, p in enumerate(ps)] for b, ps in enumerate(preds)]

    class It:
        def __init__(self):
            self.k = 0

        def __len__(self):
            return len(batches)

        def next(self):                                           # the reference calls iterator.next() (eval_acc.py:30)
            self.k += 1
            return batches[self.k - 1], [tuple(gts[self.k - 1])]

        __next__ = next

    class Loader:
        def __iter__(self):
            return It()

    metric = TextAccuracy(charset_path=None, case_sensitive=False, model_eval="vision")
    with torch.no_grad():
        res = metric.compute(torch.nn.DataParallel(model), Loader())
    out = {"pred": np.array([p for ps in preds for p in ps]), "gt": np.array([t for ts in gts for t in ts]),
           "names": np.array(list(res.keys())), "values": np.array([float(v) for v in res.values()]),
           "image_stat": np.stack([stat(b) for b in batches])}
    np.savez_compressed(os.path.join(GOLD, "eval_acc.npz"), **out)
    print("eval_acc.npz written:", {k: round(float(v), 4) for k, v in res.items() if k != "time"}, out["pred"][:4], out["gt"][:4])


def word_image(rs, h, w, n_chars, fg, bg, noise, blur=True):
    """A synthetic gray word image: n_chars blocky glyphs of gray level fg on bg, box-blurred edges + gaussian noise."""
    img = np.full((h, w), float(bg))
    cw = max(2, (w - 4) // max(n_chars, 1))
    for c in range(n_chars):
        x0 = 2 + c * cw + rs.randint(0, max(1, cw // 4))
        gw = max(1, int(cw * rs.uniform(0.4, 0.8)))
        y0 = rs.randint(1, max(2, h // 4))
        gh = max(1, int(h * rs.uniform(0.5, 0.75)))
        img[y0:y0 + gh, x0:x0 + gw] = fg
        if gw > 3 and gh > 4 and rs.uniform() < 0.5:          # a hole, so that glyphs are not convex
            img[y0 + gh // 3:y0 + 2 * gh // 3, x0 + 1:x0 + gw - 1] = bg
    if blur:
        pad = np.pad(img, 1, mode="edge")
        img = sum(pad[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)) / 9.0
    img = img + rs.normal(0, noise, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def gen_kmeans():
    """Fixtures of `clusterpixels(im, 2)` (Dino/utils/kmeans.py:7-23 = mask_create/generate_mask.py:13-29): the REAL function
    (scipy.cluster.vq.kmeans with its own random restarts) on synthetic word images, run 3 times per image with different
    numpy seeds - an image is kept only when the three runs agree (the reference itself is deterministic there)."""
    np.float = float                       # the reference was written for numpy < 1.24 (`im.astype(np.float)`)
    from Dino.utils.kmeans import clusterpixels
    rs = np.random.RandomState(7)
    imgs, masks, dropped = [], [], 0
    shapes = [(32, 100), (31, 97), (48, 160), (20, 64), (64, 256), (17, 33), (40, 40), (7, 9)]
    for i in range(48):
        h, w = shapes[i % len(shapes)]
        dark_text = rs.uniform() < 0.5
        fg, bg = (rs.randint(10, 90), rs.randint(150, 245)) if dark_text else (rs.randint(160, 250), rs.randint(5, 100))
        img = word_image(rs, h, w, rs.randint(1, 9), fg, bg, noise=rs.uniform(0, 12))
        outs = []
        for seed in (0, 1, 2):
            np.random.seed(seed)
            outs.append(np.asarray(clusterpixels(img, 2)).astype(np.uint8))
        if not all((o == outs[0]).all() for o in outs[1:]):
            dropped += 1
            continue
        imgs.append(img); masks.append(outs[0])
    flat_i = np.concatenate([a.reshape(-1) for a in imgs]); flat_m = np.concatenate([a.reshape(-1) for a in masks])
    np.savez_compressed(os.path.join(GOLD, "kmeans_masks.npz"), gray=flat_i, mask=flat_m,
                        hw=np.array([a.shape for a in imgs], dtype=np.int32))
    print(f"kmeans_masks.npz written: {len(imgs)} images ({dropped} dropped: reference output depends on its random restarts)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    os.chdir("/root/reference")  # Config() and friends use relative paths; we never write here
    torch.set_num_threads(8)
    todo = [a.only] if a.only else ["sched", "ccl", "tiny", "small", "small3", "arch", "keys", "finetune", "kmeans", "eval"]
    for t in todo:
        {"sched": gen_sched, "ccl": gen_ccl, "tiny": gen_tiny, "arch": gen_arch, "small": gen_small, "small3": gen_small3, "keys": gen_keys,
         "finetune": gen_finetune, "kmeans": gen_kmeans, "eval": gen_eval, "small_noise": gen_small_noise}[t]()
