"""Fused per-tensor-clip + AdamW + teacher EMA over the flat parameter arena (train.py:131-133, 244-272).

`FusedClipAdamW` looks like a torch optimizer where train.py touches it (param_groups[i]['lr'/'weight_decay'],
zero_grad, step, state_dict/load_state_dict in torch.optim.AdamW's layout) but a step is three HIP kernel launches
(ccd_amd/csrc/kernels/optim.h) instead of ~190 host-synchronising clips plus ~600 small AdamW kernels.
"""
from __future__ import annotations

import math

import torch

from . import ops
from .arena import ParamArena


class FusedClipAdamW:
    def __init__(self, arena: ParamArena, betas=(0.9, 0.999), eps=1e-8, clip_grad=0.0, lr=1e-3, weight_decay=1e-2,
                 global_norm=False):
        """clip_grad: 0 = off.  global_norm=False: every tensor clipped on its OWN L2 norm (clip_gradients,
        modules/utils.py:132-141, the pretraining rule); True: torch.nn.utils.clip_grad_norm_ over all tensors
        (train_finetune.py:281-282)."""
        self.arena = arena
        self.betas, self.eps, self.clip_grad, self.global_norm = betas, eps, clip_grad, global_norm
        names = list(arena.segments)
        decayed = [n for n in names if arena.params[n].requires_grad and not (n.endswith(".bias") or arena.params[n].dim() == 1)]
        plain = [n for n in names if arena.params[n].requires_grad and (n.endswith(".bias") or arena.params[n].dim() == 1)]
        # same two groups as get_params_groups (modules/utils.py:643-654); train.py overwrites lr / wd every iteration
        self.param_groups = [{"names": decayed, "params": [arena.params[n] for n in decayed], "lr": lr,
                              "weight_decay": weight_decay},
                             {"names": plain, "params": [arena.params[n] for n in plain], "lr": lr, "weight_decay": 0.0}]
        self._group_of = {n: gi for gi, g in enumerate(self.param_groups) for n in g["names"]}
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.steps = {n: 0 for n in names}
        self.never_used = set()            # tensors that have never received a gradient (conv_mla.*, cls_token)
        self._norm2 = torch.zeros(len(names), dtype=torch.float32, device=arena.device)
        # Per-tensor hyper-parameters travel host -> device as one small async copy per step.  The host runs ahead of the
        # stream (train.py reads the loss one iteration late), so a single pinned staging buffer would be overwritten by
        # step N+1 before the DMA of step N has read it: a ring of staging buffers, each guarded by the event recorded
        # behind its copy (waited for - normally long complete - before the buffer is refilled).
        on_gpu = arena.device.type == "cuda"
        self._hyper_ring = [torch.zeros((len(names), 4), dtype=torch.float32).pin_memory() if on_gpu
                            else torch.zeros((len(names), 4), dtype=torch.float32) for _ in range(4 if on_gpu else 1)]
        self._hyper_events = [None] * len(self._hyper_ring)
        self._hyper_next = 0
        self._hyper_dev = torch.zeros((len(names), 4), dtype=torch.float32, device=arena.device)

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    def mark_unused(self, names):
        """Tensors that never take part in the forward pass (torch would leave their .grad None: no AdamW update)."""
        self.never_used.update(names)

    @torch.no_grad()
    def step(self):
        arena = self.arena
        b1, b2 = self.betas
        skip = arena.skip_substrings
        slot = self._hyper_next
        self._hyper_next = (slot + 1) % len(self._hyper_ring)
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()          # the copy that last read this buffer has executed
        h = self._hyper_ring[slot]
        for n, seg in arena.segments.items():
            gi = self._group_of.get(n)
            active = gi is not None and n not in self.never_used and not any(s in n for s in skip)
            if not active:
                h[seg.index, 3] = 0.0
                continue
            g = self.param_groups[gi]
            self.steps[n] += 1
            t = self.steps[n]
            h[seg.index, 0] = g["lr"] * g["weight_decay"]
            h[seg.index, 1] = g["lr"] / (1.0 - b1 ** t)
            h[seg.index, 2] = 1.0 / math.sqrt(1.0 - b2 ** t)
            h[seg.index, 3] = 1.0
        arena.skip_substrings = set()
        self._hyper_dev.copy_(h, non_blocking=True)
        if h.is_pinned():
            self._hyper_events[slot] = torch.cuda.Event()
            self._hyper_events[slot].record()
        cs, cb, cl = arena.opt_tables()
        self._norm2.zero_()
        if self.clip_grad:
            ops.seg_sumsq(arena.grad, cs, cb, cl, self._norm2)
            if self.global_norm:                   # every tensor sees the total norm: same coefficient everywhere
                self._norm2.copy_(self._norm2.sum().expand_as(self._norm2))
        ops.adamw(arena.flat, arena.grad, self.exp_avg, self.exp_avg_sq, arena.mirror, cs, cb, cl, self._hyper_dev,
                  self._norm2, float(self.clip_grad or 0.0), b1, b2, self.eps)
        arena.refresh_transposes()

    # ------------------------------------------------------------------ torch.optim.AdamW-shaped checkpoints
    def state_dict(self):
        state, idx, groups = {}, 0, []
        for g in self.param_groups:
            ids = []
            for n in g["names"]:
                seg = self.arena.segments[n]
                if self.steps[n] > 0:
                    sl = slice(seg.offset, seg.offset + seg.numel)
                    state[idx] = {"step": torch.tensor(float(self.steps[n])),
                                  "exp_avg": self.exp_avg[sl].view(seg.shape).clone(),
                                  "exp_avg_sq": self.exp_avg_sq[sl].view(seg.shape).clone()}
                ids.append(idx)
                idx += 1
            groups.append({"lr": g["lr"], "betas": self.betas, "eps": self.eps, "weight_decay": g["weight_decay"],
                           "amsgrad": False, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        idx = 0
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            g["lr"], g["weight_decay"] = sg["lr"], sg["weight_decay"]
            for n in g["names"]:
                seg = self.arena.segments[n]
                st = sd["state"].get(idx)
                if st is not None:
                    sl = slice(seg.offset, seg.offset + seg.numel)
                    self.exp_avg[sl].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[sl].copy_(st["exp_avg_sq"].reshape(-1))
                    self.steps[n] = int(float(st["step"]))
                idx += 1


@torch.no_grad()
def ema_update(student_arena: ParamArena, teacher_arena: ParamArena, momentum: float,
               prefixes=("backbone.", "head.")):
    """teacher = m*teacher + (1-m)*student over the backbone and head parameters (train.py:264-272).
    Both arenas lay a prefix's tensors out identically, so each prefix is one contiguous range."""
    for pre in prefixes:
        slo, shi = student_arena.range_of(pre)
        tlo, thi = teacher_arena.range_of(pre)
        assert shi - slo == thi - tlo, pre
        ops.ema(teacher_arena.flat[tlo:thi], student_arena.flat[slo:shi], teacher_arena.mirror[tlo:thi], momentum)
