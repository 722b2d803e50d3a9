"""Segmentation head of the student (Dino/modules/segmentor.py: Conv_MLA 6-35, MLAHead 38-70, SegHead 73-95).

Same parameter tree / construction order as the reference (incl. the never-executed conv_mla branch, which is why 18
of its tensors never receive gradients).  The nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d children only HOLD the
parameters and running statistics (state-dict compatible, convertible by nn.SyncBatchNorm.convert_sync_batchnorm);
the computation is ccd_amd.seghead.SegHeadFn: implicit-GEMM convolutions on the MFMA GEMM kernel, fused BatchNorm+ReLU
passes and a VALU classifier conv, all hand-written HIP.  There is no library (MIOpen) or CPU path.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _conv_bn_relu(cin, cout, k):
    return [nn.Conv2d(cin, cout, k, padding=k // 2, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]


class Conv_MLA(nn.Module):
    """Constructed for state-dict / RNG compatibility; SegHead.forward never calls it (segmentor.py:80,90-95)."""

    def __init__(self, in_channels=1024, mla_channels=256):
        super().__init__()
        for n in ("mla_p2_1x1", "mla_p3_1x1", "mla_p4_1x1"):
            setattr(self, n, nn.Sequential(*_conv_bn_relu(in_channels, mla_channels, 1)))
        for n in ("mla_p2", "mla_p3", "mla_p4"):
            setattr(self, n, nn.Sequential(*_conv_bn_relu(mla_channels, mla_channels, 3)))


class MLAHead(nn.Module):
    def __init__(self, in_channels=384, mla_channels=128, mlahead_channels=64):
        super().__init__()
        for n in ("head2", "head3", "head4"):
            setattr(self, n, nn.Sequential(*_conv_bn_relu(in_channels, mla_channels, 3),
                                           *_conv_bn_relu(mla_channels, mlahead_channels, 1)))

    def forward(self, p2, p3, p4):
        return torch.cat([self.head2(p2), self.head3(p3), self.head4(p4)], dim=1)


class SegHead(nn.Module):
    def __init__(self, in_channels=384, mla_channels=128, mlahead_channels=64, num_classes=2, **kwargs):
        super().__init__(**kwargs)
        self.num_classes = num_classes
        self.conv_mla = Conv_MLA(in_channels, mla_channels)
        self.mlahead = MLAHead(in_channels=in_channels, mla_channels=mla_channels, mlahead_channels=mlahead_channels)
        self.unpool1 = nn.Sequential(nn.ConvTranspose2d(3 * mlahead_channels, 128, (4, 4), (2, 2), (1, 1)),
                                     nn.BatchNorm2d(128), nn.ReLU(True))
        self.unpool2 = nn.Sequential(nn.ConvTranspose2d(128, 128, (4, 4), (2, 2), (1, 1)), nn.BatchNorm2d(128),
                                     nn.ReLU(True))
        self.cls = nn.Conv2d(128, self.num_classes, 3, padding=1)

    def forward(self, inputs):
        """inputs: three [N,E,8,32] feature maps (channels-last views of the bf16 taps) -> fp32 logits [N,2,32,128]."""
        from .. import _lib
        from ..seghead import seg_head_forward
        if inputs[0].device.type != "cuda" and _lib._stream_override is None:
            raise RuntimeError("ccd_amd.SegHead runs on an AMD GPU only; there is no CPU path")
        n, e = inputs[0].shape[0], inputs[0].shape[1]
        assert tuple(inputs[0].shape[2:]) == (8, 32), "SegHead expects the 8x32 token grid of a 32x128 crop"
        taps = [t.permute(0, 2, 3, 1).reshape(n * 256, e).to(torch.bfloat16) for t in inputs]   # views for bf16 taps
        return seg_head_forward(self, taps, n)
