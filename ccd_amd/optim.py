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


class HostStaging:
    """A small host table that travels to a device tensor once per step as one async copy.  The host runs ahead of the stream
    (train.py reads the loss one iteration late), so a single pinned staging buffer would be overwritten by step N+1 before the
    DMA of step N has read it: a ring of staging buffers, each guarded by the event recorded behind its copy (waited for -
    normally long complete - before the buffer is refilled)."""

    def __init__(self, shape, dtype, device, depth=4):
        on_gpu = torch.device(device).type == "cuda"
        self.ring = [torch.zeros(shape, dtype=dtype).pin_memory() if on_gpu else torch.zeros(shape, dtype=dtype)
                     for _ in range(depth if on_gpu else 1)]
        self.events = [None] * len(self.ring)
        self.next = 0
        self.dev = torch.zeros(shape, dtype=dtype, device=device)

    def begin(self):
        """-> the host buffer to fill for this step."""
        self.slot = self.next
        self.next = (self.slot + 1) % len(self.ring)
        if self.events[self.slot] is not None:
            self.events[self.slot].synchronize()            # the copy that last read this buffer has executed
        return self.ring[self.slot]

    def commit(self):
        h = self.ring[self.slot]
        self.dev.copy_(h, non_blocking=True)
        if h.is_pinned():
            self.events[self.slot] = torch.cuda.Event()
            self.events[self.slot].record()
        return self.dev


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
        # per-tensor hyper-parameters: host -> device as one small async copy per step
        self._hyper = HostStaging((len(names), 4), torch.float32, arena.device)
        self._hyper_dev = self._hyper.dev

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    def mark_unused(self, names):
        """Tensors that never take part in the forward pass (torch would leave their .grad None: no AdamW update)."""
        self.never_used.update(names)

    @torch.no_grad()
    def step(self):
        self.stage_hyper()
        self.launch_step()

    @torch.no_grad()
    def stage_hyper(self):
        """The host half of a step: step counters, bias corrections, lr / wd and the frozen-tensor switches of THIS iteration
        go to the device table (an eager copy - a HIP graph of the step replays `launch_step` only)."""
        arena = self.arena
        b1, b2 = self.betas
        skip = arena.skip_substrings
        h = self._hyper.begin()
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
        self._hyper.commit()

    @torch.no_grad()
    def launch_step(self):
        """The device half: per-tensor (or global) clip + AdamW + mirror refresh; reads the staged table."""
        arena = self.arena
        b1, b2 = self.betas
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
               prefixes=("backbone.", "head."), d_m=None):
    """teacher = m*teacher + (1-m)*student over the backbone and head parameters (train.py:264-272).
    Both arenas lay a prefix's tensors out identically, so each prefix is one contiguous range.
    d_m (fp32 [2] on the device, {m, 1 - m}): the momentum is read when the kernel RUNS (replays of a graphed step)."""
    for pre in prefixes:
        slo, shi = student_arena.range_of(pre)
        tlo, thi = teacher_arena.range_of(pre)
        assert shi - slo == thi - tlo, pre
        ops.ema(teacher_arena.flat[tlo:thi], student_arena.flat[slo:shi], teacher_arena.mirror[tlo:thi], momentum, d_m)
