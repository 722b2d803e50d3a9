"""DINOLoss: segmentation CE + character-to-character distillation CE + teacher centre EMA
(Dino/loss/Dino_loss.py:35-143), computed by the fused HIP kernels of ccd_amd/csrc/kernels/loss.h."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import engine, ops


class DINOLoss(nn.Module):
    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs,
                 student_temp=0.1, center_momentum=0.9):
        super().__init__()
        if ncrops != 2:
            raise NotImplementedError("the CCD pretraining path uses exactly two views (crops_number: 2)")
        self.student_temp, self.center_momentum, self.ncrops = student_temp, center_momentum, ncrops
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.teacher_temp_schedule = np.concatenate((
            np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
            np.ones(nepochs - warmup_teacher_temp_epochs) * teacher_temp))
        self.losses = {}

    @property
    def last_losses(self):
        return self.losses

    @staticmethod
    def _logits_and_count(out):
        raw = getattr(out, "raw", None)
        if raw is not None and raw("logits_buf") is not None:
            return raw("logits_buf"), raw("selection").total, True
        t = out["instances_view"].contiguous().float()
        return t, torch.full((1,), t.shape[0] // 2, dtype=torch.int32, device=t.device), False

    def forward(self, student_output, teacher_output, epoch):
        self.losses = {}
        s_logits, d_total, direct = self._logits_and_count(student_output)
        t_logits, _, _ = self._logits_and_count(teacher_output)
        # --- segmentation loss against [masks, warped masks] (train.py:234-237 builds 'gt')
        gt = student_output["gt"]
        mask_a = gt[0].contiguous().float()
        gt_b = gt[1]
        idmap_b = gt_b if gt_b.dtype == torch.uint8 else ops.mask_to_idmap(gt_b.contiguous().float())
        mask_loss = engine.SegLossFn.apply(student_output["mask"], mask_a, idmap_b)[0]
        # --- character-to-character distillation
        temp = float(self.teacher_temp_schedule[epoch])
        center = self.center.view(-1)
        dino_loss = engine.dino_loss(s_logits, t_logits.detach(), center, d_total, self.student_temp, temp, direct)[0]
        self.update_center(t_logits.detach(), d_total)
        self.losses["mask_loss"] = mask_loss
        self.losses["Dino_loss"] = dino_loss
        return mask_loss + dino_loss

    @torch.no_grad()
    def update_center(self, teacher_logits, d_total=None):
        """sum over local rows -> all_reduce(SUM) -> / (local rows * world) -> EMA (Dino_loss.py:133-143)."""
        if d_total is None:
            d_total = torch.full((1,), teacher_logits.shape[0] // 2, dtype=torch.int32, device=teacher_logits.device)
        batch_sum = torch.zeros(self.center.shape[1], dtype=torch.float32, device=teacher_logits.device)
        engine.logit_column_sums(teacher_logits, d_total, batch_sum, rows_mul=2)      # (a matrix-vector product where the head left its factors)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(batch_sum)
            world = dist.get_world_size()
        ops.center_ema(self.center.view(-1), batch_sum, d_total, world, self.center_momentum)
