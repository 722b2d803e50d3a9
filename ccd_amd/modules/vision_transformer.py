"""MI355X-native ViT backbone + DINO head behind the reference's module surface
(Dino/modules/vision_transformer.py: VisionTransformer 134-251, vit_tiny/small/base 273-291, DINOHead 294-328).

The nn.Module tree only CARRIES parameters - same names, shapes, construction order and init RNG stream as the
reference, so state dicts and seeds are interchangeable - while forward()/backward run the HIP kernels of
ccd_amd.engine on the flat parameter arena.  There is no PyTorch-eager fallback.
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from .. import _lib, engine
from ..arena import ParamArena
from .utils import trunc_normal_

__all__ = ["VisionTransformer", "DINOHead", "vit_tiny", "vit_small", "vit_base"]


def _holder(**children) -> nn.Module:
    """A bare nn.Module that just names its children (state-dict paths follow the reference)."""
    m = nn.Module()
    for k, v in children.items():
        m.add_module(k, v)
    return m


class ArenaModule(nn.Module):
    """Mixin: parameters live in a ccd_amd.arena.ParamArena, attached lazily once the module sits on a GPU."""

    arena: ParamArena | None = None
    arena_prefix: str = ""
    grad_ready_hook = None           # set by ccd_amd.parallel to learn when a prefix's gradients are final

    def _transposed_names(self):     # 2-D weights whose backward needs a bf16 W^T mirror
        return []

    def attach_arena(self, arena: ParamArena, prefix: str):
        self.arena, self.arena_prefix = arena, prefix

    def ensure_arena(self):
        if self.arena is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda" and _lib._stream_override is None:
                raise RuntimeError("ccd_amd modules run on an AMD GPU only (move the model with .cuda() first); "
                                   "there is no CPU path")
            arena = ParamArena(list(self.named_parameters()), dev, with_grad=True,
                               transposed=self._transposed_names())
            self.attach_arena(arena, "")
        if self.arena.stale:
            self.arena.refresh_mirrors()
        return self.arena

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        if self.arena is not None:
            self.arena.stale = True      # values were copied into the fp32 views; bf16 mirrors must follow


class VisionTransformer(ArenaModule):
    """Dual-view ViT encoder for 32x128 crops: forward(x[N,3,32,128]) -> (tokens [N,256,E], [3 x (N,E,8,32)])."""

    def __init__(self, img_size=[32, 128], patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, out_indices=[2, 4, 6], **kwargs):
        super().__init__()
        if in_chans != 3 or patch_size != 4 or list(img_size) != [32, 128] or embed_dim % num_heads or \
                embed_dim // num_heads != 64 or mlp_ratio != 4 or not qkv_bias or drop_rate or attn_drop_rate:
            raise NotImplementedError("HIP kernels cover the CCD pretraining shapes: 32x128 input, patch 4, head_dim 64, "
                                      "mlp_ratio 4, qkv_bias, no dropout")
        E = embed_dim
        self.num_features = self.embed_dim = E
        self.out_indices = list(out_indices)
        hidden = int(E * mlp_ratio)
        # construction order == reference order (init RNG stream, vision_transformer.py:141-166)
        self.patch_embed = _holder(proj=nn.Conv2d(in_chans, E, kernel_size=patch_size, stride=patch_size))
        self.patch_embed.patch_size = patch_size
        self.patch_embed.num_patches = (img_size[0] // patch_size) * (img_size[1] // patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, E))          # never enters the sequence (:229-231)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, E))
        self.blocks = nn.ModuleList([
            _holder(norm1=norm_layer(E), attn=_holder(qkv=nn.Linear(E, 3 * E, bias=qkv_bias), proj=nn.Linear(E, E)),
                    norm2=norm_layer(E), mlp=_holder(fc1=nn.Linear(E, hidden), fc2=nn.Linear(hidden, E)))
            for _ in range(depth)])
        self.norm = norm_layer(E)
        self.head = nn.Linear(E, num_classes) if num_classes > 0 else nn.Identity()
        self.norm_seg = nn.Sequential(norm_layer(E), norm_layer(E), norm_layer(E))
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():      # same traversal order as nn.Module.apply for Linear layers
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        eps = getattr(self.norm, "eps", 1e-6)
        self.spec = engine.VitSpec(E, depth, num_heads, self.out_indices, patch_size, eps, drop_path_rate)
        self.register_buffer("resample", engine.bicubic_resample_matrix(int(math.sqrt(self.patch_embed.num_patches)),
                                                                        img_size[0] // patch_size,
                                                                        img_size[1] // patch_size), persistent=False)

    def _transposed_names(self):
        names = []
        for i in range(self.spec.depth):
            names += [f"blocks.{i}.attn.qkv.weight", f"blocks.{i}.attn.proj.weight", f"blocks.{i}.mlp.fc1.weight",
                      f"blocks.{i}.mlp.fc2.weight"]
        return names

    def tokens_and_taps(self, x, need_taps=True):
        """bf16 token tensors in the kernels' native [N,256,E] layout (what the rest of ccd_amd consumes).
        need_taps=False skips the three norm_seg outputs (the teacher never feeds a segmentation head)."""
        self.ensure_arena()
        anchor = self.arena.params[self.arena_prefix + "pos_embed"]
        return engine.BackboneFn.apply(anchor, x.contiguous().float(), self, need_taps)

    def to_2D(self, t):
        return t.reshape(t.shape[0], 8, 32, -1).permute(0, 3, 1, 2)

    def forward(self, x):
        tokens, *taps = self.tokens_and_taps(x)
        return tokens, [self.to_2D(t) for t in taps]


def vit_tiny(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_small(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_base(patch_size=16, **kwargs):
    # the reference's "base" is E=512 / 8 heads (vision_transformer.py:287-291), not 768/12
    return VisionTransformer(patch_size=patch_size, embed_dim=512, depth=12, num_heads=8, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_base_768(patch_size=16, **kwargs):
    """E = 768 / 12 heads: the shape BASELINE.json config #4 names for "ViT-Base".  The reference's own `vit_base` is 512 / 8
    (above); 768 / 12 is what its VisionTransformer constructor defaults to (vision_transformer.py:117-120).  Unfused path
    (the row-owner kernels stop at E = 512): tiled GEMMs, stand-alone LayerNorm kernels with 3 of 4 column steps live."""
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


class DINOHead(ArenaModule):
    """3-layer GELU MLP -> L2 normalise -> weight-normalised Linear(bottleneck, out_dim, bias=False)."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=2048,
                 bottleneck_dim=256):
        super().__init__()
        if use_bn or nlayers != 3:
            raise NotImplementedError("HIP DINOHead covers the shipped configuration: nlayers=3, use_bn=False")
        # ccd_weightnorm_fwd writes the transposed bf16 copy of the last layer two output units (4 bytes) at a time and reads a row
        # of v as 16-byte pieces: say so here, at model build time, not as a CCD_ESHAPE from the first forward pass
        if out_dim % 2 or bottleneck_dim % 4 or bottleneck_dim > 1024:
            raise ValueError(f"HIP DINOHead: out_dim must be even and bottleneck_dim a multiple of 4 (<= 1024); got out_dim={out_dim}, "
                             f"bottleneck_dim={bottleneck_dim}")
        self.mlp = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
                                 nn.Linear(hidden_dim, bottleneck_dim))
        for m in self.mlp:
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                nn.init.constant_(m.bias, 0)
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, out_dim, bias=False))
        self.last_layer.weight_g.data.fill_(1)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False
        self.out_dim = out_dim

    @property
    def weight_g_trainable(self):
        return self.last_layer.weight_g.requires_grad

    def _transposed_names(self):
        return ["mlp.0.weight", "mlp.2.weight", "mlp.4.weight"]

    def forward_rows(self, rows, d_total, lazy=False):
        """rows bf16 [max_rows, in_dim] with the live row count 2*d_total[0] on the device -> logits fp32; `lazy`: an
        engine.LazyLogits where the fused head + loss kernels take the shape (the [max_rows, out_dim] matrix is never written)."""
        self.ensure_arena()
        return engine.head_rows(self, rows, d_total, 2, lazy)

    def forward(self, x):
        self.ensure_arena()
        x2 = x.reshape(-1, x.shape[-1]).to(torch.bfloat16).contiguous()
        d_total = torch.full((1,), x2.shape[0], dtype=torch.int32, device=x2.device)
        out = engine.HeadFn.apply(x2, self, d_total, 1)
        return out.reshape(*x.shape[:-1], -1)
