"""One CCD finetune iteration on the MI355X-native modules (the body of train_finetune.py:262-289), shared by
train_finetune.py, bench.py and the parity tests."""
from __future__ import annotations

import torch

from .model.dino_vision import DINO_Finetune
from .optim import FusedClipAdamW


class FinetuneConfig:
    """The attributes DINO_Finetune reads from the reference's Config (Dino/configs/CCD_vision_model_ARD.yaml)."""

    def __init__(self, arch="vit_small", patch_size=4, drop_path_rate=0.1, decoder_n_layers=6, decoder_d_embedding=512,
                 decoder_n_head=8, decoder_d_model=512, decoder_d_inner=256, decoder_d_k=64, decoder_d_v=64,
                 decoder_max_seq_len=25):
        self.arch, self.patch_size, self.drop_path_rate = arch, patch_size, drop_path_rate
        self.decoder_n_layers, self.decoder_d_embedding, self.decoder_n_head = decoder_n_layers, decoder_d_embedding, decoder_n_head
        self.decoder_d_model, self.decoder_d_inner, self.decoder_d_k, self.decoder_d_v = decoder_d_model, decoder_d_inner, decoder_d_k, decoder_d_v
        self.decoder_max_seq_len = decoder_max_seq_len


def build_model(config=None, device="cuda", dropout=None):
    """DINO_Finetune on `device` with its arena attached; dropout=0.0 switches every nn.Dropout off (parity runs)."""
    model = DINO_Finetune(config or FinetuneConfig()).to(device)
    if dropout is not None:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = float(dropout)
    model.ensure_arena()
    model.train()
    return model


def make_optimizer(model, lr=0.0005, weight_decay=0.05, clip_grad=None):
    """torch.optim.AdamW(get_params_groups(model), lr, betas=(0.9, 0.999), weight_decay) (train_finetune.py:221-224);
    clip_grad (train_finetune.py:281-282) is torch.nn.utils.clip_grad_norm_: ONE norm over all gradients."""
    mod = model.module if hasattr(model, "module") else model
    opt = FusedClipAdamW(mod.ensure_arena(), clip_grad=float(clip_grad or 0.0), lr=lr, weight_decay=weight_decay,
                         global_norm=True)
    opt.mark_unused(mod.unused_parameter_names())
    return opt


def training_iteration(model, optimizer: FusedClipAdamW, images, labels, lr):
    """images [B,3,32,128], labels int64 [B,T] (AttnConvertor.str2tensor) -> loss (device tensor)."""
    for g in optimizer.param_groups:
        g["lr"] = float(lr)
    losses, attn = model(images, labels, return_loss=True)
    loss = losses.mean()
    optimizer.zero_grad()
    loss.backward()
    if hasattr(model, "finish_gradient_sync"):
        model.finish_gradient_sync()
    optimizer.step()
    return loss.detach(), attn
