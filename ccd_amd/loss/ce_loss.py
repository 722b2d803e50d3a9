"""TFLoss of the recognition head (Dino/loss/ce_loss.py:3-128) on the fused HIP cross-entropy kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import finetune_engine as fe


class TFLoss(nn.Module):
    """CrossEntropy of logits[:, :-1] against padded_targets[:, 1:] (the first target is always <SOS>), `ignore_index`
    rows skipped, mean over the rest (reduction='mean', flatten=True: the only configuration the path uses)."""

    def __init__(self, ignore_index=-1, reduction='mean', flatten=True, **kwargs):
        super().__init__()
        assert isinstance(ignore_index, int) and isinstance(flatten, bool)
        if reduction != 'mean':
            raise NotImplementedError("HIP TFLoss implements reduction='mean' (DINO_Finetune's setting)")
        self.ignore_index, self.flatten = ignore_index, flatten

    def forward(self, outputs, targets_dict, img_metas=None):
        targets = targets_dict['padded_targets'].to(outputs.device).long().contiguous()
        if outputs.dtype != torch.float32 or outputs.stride(-1) != 1:
            outputs = outputs.float().contiguous()
        return fe.TFLossFn.apply(outputs, targets, self.ignore_index)
