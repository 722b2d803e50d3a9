"""One CCD pretraining iteration on the MI355X-native modules (the body of train.py:221-272), shared by train.py,
bench.py, __graft_entry__.smoke() and the parity tests."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from .loss.Dino_loss import DINOLoss
from .model.dino_vision import ABIDINOModel
from .modules import vision_transformer as vits
from .modules.segmentor import SegHead
from .optim import FusedClipAdamW, ema_update


def build_networks(arch="vit_small", patch_size=4, out_dim=65536, drop_path_rate=0.1, norm_last_layer=False,
                   use_bn_in_head=False, seg_channel=None, backbone_kwargs=None, head_kwargs=None, device="cuda"):
    """Student / teacher exactly in train.py:63-114's construction order (same RNG stream as the reference)."""
    bk, hk = backbone_kwargs or {}, head_kwargs or {}
    if arch in vits.__dict__:
        student_b = vits.__dict__[arch](patch_size=patch_size, drop_path_rate=drop_path_rate, **bk)
        teacher_b = vits.__dict__[arch](patch_size=patch_size, **bk)
    else:   # explicit dimensions (tests / tiny models)
        mk = lambda **kw: vits.VisionTransformer(patch_size=patch_size, qkv_bias=True, mlp_ratio=4,
                                                 norm_layer=lambda e: nn.LayerNorm(e, eps=1e-6), **bk, **kw)
        student_b, teacher_b = mk(drop_path_rate=drop_path_rate), mk()
    E = student_b.embed_dim
    student = ABIDINOModel(student_b, SegHead(in_channels=seg_channel or E, mla_channels=128, mlahead_channels=64,
                                              num_classes=2),
                           vits.DINOHead(E, out_dim, use_bn=use_bn_in_head, norm_last_layer=norm_last_layer, **hk))
    teacher = ABIDINOModel(teacher_b, None, vits.DINOHead(E, out_dim, use_bn_in_head, **hk))
    student, teacher = student.to(device), teacher.to(device)
    student.ensure_arena()
    teacher.ensure_arena()
    teacher.backbone.load_state_dict(student.backbone.state_dict())
    teacher.head.load_state_dict(student.head.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    teacher.ensure_arena()          # refreshes the bf16 mirrors after the state-dict copy
    return student, teacher


def make_optimizer(student_module, clip_grad=3.0):
    arena = student_module.ensure_arena()
    opt = FusedClipAdamW(arena, clip_grad=clip_grad)
    opt.mark_unused(student_module.unused_parameter_names())
    return opt


def training_iteration(student, teacher, dino_loss: DINOLoss, optimizer: FusedClipAdamW, images, masks, metrics,
                       epoch, lr, wd, momentum, freeze_last_layer=1, check_finite=False):
    """student / teacher may be wrapped (ccd_amd.parallel.DataParallel); returns the loss as a device tensor."""
    s_mod = student.module if hasattr(student, "module") else student
    t_mod = teacher.module if hasattr(teacher, "module") else teacher
    for i, g in enumerate(optimizer.param_groups):
        g["lr"] = float(lr)
        if i == 0:
            g["weight_decay"] = float(wd)
    metrics = metrics.float()
    s_out = student(images, metrics, masks, epoch, clusters=None)
    with torch.no_grad():
        t_out = teacher(images, metrics, None, None, clusters=s_out["zero"], index=None)
    # gt = [masks, warped masks > 0.1]  (train.py:234-237); the warped half stays an id map on the device
    masks_image = ops.warp_idmap(ops.mask_to_idmap(masks.contiguous().float()), metrics.contiguous())
    s_out["gt"] = [masks, masks_image]
    loss = dino_loss(s_out, t_out, epoch)
    if check_finite and not math.isfinite(loss.item()):
        raise FloatingPointError("Loss is {}, stopping training".format(loss.item()))
    optimizer.zero_grad()
    loss.backward()
    if hasattr(student, "finish_gradient_sync"):
        student.finish_gradient_sync()
    if epoch < freeze_last_layer:
        s_mod.arena.skip_substrings.add("last_layer")        # cancel_gradients_last_layer (modules/utils.py:144-149)
    optimizer.step()                                         # per-tensor clip + AdamW, fused
    ema_update(s_mod.arena, t_mod.arena, float(momentum))
    return loss.detach()
