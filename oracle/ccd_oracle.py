"""ORACLE - CPU restatement (PyTorch fp32 + numpy) of the CCD pretraining step.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
path (ccd_amd/) never does and fails loudly without its HIP library.

The reference is 100 % Python/PyTorch, so the restatement is a *functional* PyTorch
program over a flat {state-dict key: tensor} table with the reference's exact key names.  Each function
cites the reference lines it follows.  It is pinned (tests/test_oracle_golden.py) against fixtures that
tools/gen_golden.py produced by running the real reference in the build container:
    tests/golden/{sched,ccl_cases,tiny_step,small_step}.npz, state_keys.json
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ccl_np


# ----------------------------------------------------------------------------------------------- specs
@dataclass
class Spec:
    embed_dim: int = 384
    depth: int = 12
    heads: int = 6
    patch: int = 4
    taps: tuple = (2, 4, 6)              # vision_transformer.py:139 out_indices
    out_dim: int = 65536
    head_hidden: int = 2048
    head_bottleneck: int = 256
    norm_last_layer: bool = True         # student: config.norm_last_layer, teacher: default True (train.py:87-91)
    seg_in: int = 384
    drop_path_rate: float = 0.0
    img_h: int = 32
    img_w: int = 128
    ln_eps: float = 1e-6                 # vision_transformer.py:276

    @property
    def grid(self):
        return self.img_h // self.patch, self.img_w // self.patch

    @property
    def tokens(self):
        g = self.grid
        return g[0] * g[1]


ARCH = {  # vision_transformer.py:273-291
    "vit_tiny": dict(embed_dim=192, depth=12, heads=3),
    "vit_small": dict(embed_dim=384, depth=12, heads=6),
    "vit_base": dict(embed_dim=512, depth=12, heads=8),
    "vit_base_768": dict(embed_dim=768, depth=12, heads=12),     # the constructor's defaults (vision_transformer.py:117-120); BASELINE config #4's shape
}


# ------------------------------------------------------------------------------------------------ init
def _trunc_normal(t: torch.Tensor, std: float):
    """modules/utils.py:523-561 with mean 0, a=-2, b=2 (the only way the path calls it)."""
    cdf = lambda v: (1.0 + math.erf(v / math.sqrt(2.0))) / 2.0
    lo, hi = cdf(-2.0 / std), cdf(2.0 / std)
    with torch.no_grad():
        t.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.0)).add_(0.0).clamp_(min=-2.0, max=2.0)


def _bn_entries(table, key, ch):
    bn = nn.BatchNorm2d(ch)
    for k, v in bn.state_dict().items():
        table[f"{key}.{k}"] = v.clone()


def _init_backbone(spec: Spec) -> OrderedDict:
    """Same RNG consumption order as VisionTransformer.__init__ (vision_transformer.py:136-180)."""
    E = spec.embed_dim
    conv = nn.Conv2d(3, E, spec.patch, spec.patch)
    lin = []
    for _ in range(spec.depth):
        lin.append([nn.Linear(E, 3 * E), nn.Linear(E, E), nn.Linear(E, 4 * E), nn.Linear(4 * E, E)])
    cls_token = torch.zeros(1, 1, E)
    pos = torch.zeros(1, spec.tokens, E)
    _trunc_normal(pos, 0.02)
    _trunc_normal(cls_token, 0.02)
    for blk in lin:
        for l in blk:
            _trunc_normal(l.weight, 0.02)
            nn.init.zeros_(l.bias)
    t = OrderedDict()
    t["cls_token"], t["pos_embed"] = cls_token, pos
    t["patch_embed.proj.weight"], t["patch_embed.proj.bias"] = conv.weight.detach(), conv.bias.detach()
    for i, (qkv, proj, fc1, fc2) in enumerate(lin):
        b = f"blocks.{i}."
        t[b + "norm1.weight"], t[b + "norm1.bias"] = torch.ones(E), torch.zeros(E)
        t[b + "attn.qkv.weight"], t[b + "attn.qkv.bias"] = qkv.weight.detach(), qkv.bias.detach()
        t[b + "attn.proj.weight"], t[b + "attn.proj.bias"] = proj.weight.detach(), proj.bias.detach()
        t[b + "norm2.weight"], t[b + "norm2.bias"] = torch.ones(E), torch.zeros(E)
        t[b + "mlp.fc1.weight"], t[b + "mlp.fc1.bias"] = fc1.weight.detach(), fc1.bias.detach()
        t[b + "mlp.fc2.weight"], t[b + "mlp.fc2.bias"] = fc2.weight.detach(), fc2.bias.detach()
    t["norm.weight"], t["norm.bias"] = torch.ones(E), torch.zeros(E)
    for j in range(3):
        t[f"norm_seg.{j}.weight"], t[f"norm_seg.{j}.bias"] = torch.ones(E), torch.zeros(E)
    return t


def _init_seg(spec: Spec) -> OrderedDict:
    """SegHead(in, 128, 64, 2): segmentor.py:77-88, default torch inits, construction order preserved."""
    C, M, Hc = spec.seg_in, 128, 64
    t = OrderedDict()
    for n in ("mla_p2_1x1", "mla_p3_1x1", "mla_p4_1x1"):       # Conv_MLA, never executed (segmentor.py:6-22)
        t[f"conv_mla.{n}.0.weight"] = nn.Conv2d(C, M, 1, bias=False).weight.detach()
        _bn_entries(t, f"conv_mla.{n}.1", M)
    for n in ("mla_p2", "mla_p3", "mla_p4"):
        t[f"conv_mla.{n}.0.weight"] = nn.Conv2d(M, M, 3, padding=1, bias=False).weight.detach()
        _bn_entries(t, f"conv_mla.{n}.1", M)
    for n in ("head2", "head3", "head4"):                       # MLAHead segmentor.py:41-64
        t[f"mlahead.{n}.0.weight"] = nn.Conv2d(C, M, 3, padding=1, bias=False).weight.detach()
        _bn_entries(t, f"mlahead.{n}.1", M)
        t[f"mlahead.{n}.3.weight"] = nn.Conv2d(M, Hc, 1, bias=False).weight.detach()
        _bn_entries(t, f"mlahead.{n}.4", Hc)
    for n, cin in (("unpool1", 3 * Hc), ("unpool2", 128)):
        ct = nn.ConvTranspose2d(cin, 128, (4, 4), (2, 2), (1, 1))
        t[f"{n}.0.weight"], t[f"{n}.0.bias"] = ct.weight.detach(), ct.bias.detach()
        _bn_entries(t, f"{n}.1", 128)
    cls = nn.Conv2d(128, 2, 3, padding=1)
    t["cls.weight"], t["cls.bias"] = cls.weight.detach(), cls.bias.detach()
    return t


def _init_head(spec: Spec) -> OrderedDict:
    """DINOHead: vision_transformer.py:294-322 (use_bn False, nlayers 3)."""
    E, Hd, Bn, K = spec.embed_dim, spec.head_hidden, spec.head_bottleneck, spec.out_dim
    l0, l2, l4 = nn.Linear(E, Hd), nn.Linear(Hd, Hd), nn.Linear(Hd, Bn)
    for l in (l0, l2, l4):
        _trunc_normal(l.weight, 0.02)
        nn.init.zeros_(l.bias)
    last = nn.Linear(Bn, K, bias=False)
    t = OrderedDict()
    for i, l in ((0, l0), (2, l2), (4, l4)):
        t[f"mlp.{i}.weight"], t[f"mlp.{i}.bias"] = l.weight.detach(), l.bias.detach()
    t["last_layer.weight_g"] = torch.ones(K, 1)
    t["last_layer.weight_v"] = last.weight.detach()
    return t


@dataclass
class Net:
    """A network as a flat table of tensors with the reference's state-dict keys."""
    spec: Spec
    P: OrderedDict
    trainable: list = field(default_factory=list)

    def params(self):
        return [(k, self.P[k]) for k in self.trainable]


def build_pair(spec: Spec, seed: int):
    """Student + teacher exactly as train.py:63-114 builds them (construction order = RNG order)."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    sb = _init_backbone(spec)
    tb = _init_backbone(spec)              # teacher backbone consumes RNG too, then is overwritten
    seg = _init_seg(spec)
    sh = _init_head(spec)
    _ = _init_head(spec)
    S, T = OrderedDict(), OrderedDict()
    for k, v in sb.items():
        S["backbone." + k] = v.clone()
        T["backbone." + k] = v.clone()     # train.py:109-110
    for k, v in seg.items():
        S["segmentation." + k] = v.clone()
    for k, v in sh.items():
        S["head." + k] = v.clone()
        T["head." + k] = v.clone()
    del tb
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked"))
    s_train = [k for k in S if not is_buf(k)]
    if spec.norm_last_layer:
        s_train.remove("head.last_layer.weight_g")   # vision_transformer.py:315-316
    for k in s_train:
        S[k].requires_grad_(True)
    return Net(spec, S, s_train), Net(spec, T, [])


# -------------------------------------------------------------------------------------------- backbone
def resample_pos_embed(pos: torch.Tensor, spec: Spec) -> torch.Tensor:
    """interpolate_pos_encoding, vision_transformer.py:182-201: the bicubic branch always runs for 32x128
    inputs because the caller's (w, h) are (32, 128) (vision_transformer.py:226,234)."""
    n = pos.shape[1]
    side = int(math.sqrt(n))
    gh, gw = spec.grid
    sf = ((gh + 0.1) / math.sqrt(n), (gw + 0.1) / math.sqrt(n))
    grid = pos.reshape(1, side, side, -1).permute(0, 3, 1, 2)
    out = F.interpolate(grid, scale_factor=sf, mode="bicubic")
    assert out.shape[-2:] == (gh, gw)
    return out.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)


def _drop_path(y, mask_keep):
    """DropPath (vision_transformer.py:27-35) with an INJECTED per-sample 0/1 tensor (RNG parity is not possible)."""
    if mask_keep is None:
        return y
    mask, keep = mask_keep
    return y / keep * mask.view(-1, *([1] * (y.dim() - 1)))


def backbone_forward(P, pre, x, spec: Spec, drop=None):
    """VisionTransformer.forward, vision_transformer.py:240-251 -> (tokens [N,T,E], 3 x [N,E,gh,gw])."""
    E, h = spec.embed_dim, spec.heads
    d = E // h
    t = F.conv2d(x, P[pre + "patch_embed.proj.weight"], P[pre + "patch_embed.proj.bias"], stride=spec.patch)
    t = t.flatten(2).transpose(1, 2) + resample_pos_embed(P[pre + "pos_embed"], spec)
    n, T, _ = t.shape
    taps = []
    for i in range(spec.depth):
        b = f"{pre}blocks.{i}."
        y = F.layer_norm(t, (E,), P[b + "norm1.weight"], P[b + "norm1.bias"], spec.ln_eps)
        qkv = F.linear(y, P[b + "attn.qkv.weight"], P[b + "attn.qkv.bias"]).reshape(n, T, 3, h, d)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        a = torch.softmax((q @ k.transpose(-2, -1)) * d ** -0.5, dim=-1)
        o = (a @ v).transpose(1, 2).reshape(n, T, E)
        o = F.linear(o, P[b + "attn.proj.weight"], P[b + "attn.proj.bias"])
        t = t + _drop_path(o, None if drop is None else drop[i][0])
        y = F.layer_norm(t, (E,), P[b + "norm2.weight"], P[b + "norm2.bias"], spec.ln_eps)
        y = F.linear(F.gelu(F.linear(y, P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"])),
                     P[b + "mlp.fc2.weight"], P[b + "mlp.fc2.bias"])
        t = t + _drop_path(y, None if drop is None else drop[i][1])
        if i + 1 in spec.taps:
            j = len(taps)
            z = F.layer_norm(t, (E,), P[f"{pre}norm_seg.{j}.weight"], P[f"{pre}norm_seg.{j}.bias"], spec.ln_eps)
            taps.append(z.reshape(n, *spec.grid, E).permute(0, 3, 1, 2))
    t = F.layer_norm(t, (E,), P[pre + "norm.weight"], P[pre + "norm.bias"], spec.ln_eps)
    return t, taps


# -------------------------------------------------------------------------------------------- seg head
def _bn_train(P, key, x, stats_sink=None):
    """BatchNorm2d in train mode (batch statistics, momentum 0.1, eps 1e-5) with running-stat update."""
    rm, rv = P[key + ".running_mean"], P[key + ".running_var"]
    y = F.batch_norm(x, rm, rv, P[key + ".weight"], P[key + ".bias"], training=True, momentum=0.1, eps=1e-5)
    P[key + ".num_batches_tracked"] += 1
    return y


def seg_head_forward(P, pre, taps):
    """SegHead.forward, segmentor.py:90-95 (MLAHead 66-70).  conv_mla is constructed but never called."""
    branches = []
    for name, x in zip(("head2", "head3", "head4"), taps):
        k = f"{pre}mlahead.{name}"
        x = F.relu(_bn_train(P, k + ".1", F.conv2d(x, P[k + ".0.weight"], padding=1)))
        x = F.relu(_bn_train(P, k + ".4", F.conv2d(x, P[k + ".3.weight"])))
        branches.append(x)
    x = torch.cat(branches, dim=1)
    for name in ("unpool1", "unpool2"):
        k = pre + name
        x = F.conv_transpose2d(x, P[k + ".0.weight"], P[k + ".0.bias"], stride=2, padding=1)
        x = F.relu(_bn_train(P, k + ".1", x))
    return F.conv2d(x, P[pre + "cls.weight"], P[pre + "cls.bias"], padding=1)


# ------------------------------------------------------------------------------- character-region path
def label_batch(mask_np: np.ndarray) -> np.ndarray:
    """dino_vision.py:59-63 / 64-70: label every image -> uint8 id maps [B,H,W]."""
    return np.stack([ccl_np.label_idmap(m) for m in mask_np])


def warp_planes(src: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """dino_vision.py:72-77 (and train.py:234-236 for the gt mask): affine_grid + bilinear grid_sample > 0.1."""
    grid = F.affine_grid(theta[:, :2, :], size=(src.shape[0], 1, src.shape[2], src.shape[3]), align_corners=False)
    out = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    return (out > 0.1).float()


def region_pool(region_f: torch.Tensor, clusters: torch.Tensor):
    """ABIDINOModel.attention, dino_vision.py:38-49."""
    n, E, gh, gw = region_f.shape
    c = F.interpolate(clusters, size=(gh, gw), mode="bilinear", align_corners=None)
    tot = c.reshape(n, 26, -1).sum(-1)
    w = c / tot[:, :, None, None]
    w = torch.where(torch.isnan(w), torch.zeros_like(w), w)
    v = region_f.permute(0, 2, 3, 1).reshape(n, -1, E)
    return torch.bmm(w.reshape(n, 26, -1), v), tot > 0


def select_rows(vecs: torch.Tensor, index: torch.Tensor):
    """dino_vision.py:82-87: rows 0..min(len,25) of both views, len = clamp(#regions of view 1, 3, 26)."""
    b2 = index.shape[0]
    length = torch.clamp(index.sum(1), 3, 26).unsqueeze(-1)[: b2 // 2]
    new_index = torch.arange(26).unsqueeze(0) <= length
    rows = torch.cat([vecs[: b2 // 2][new_index], vecs[b2 // 2:][new_index]])
    return rows, new_index


def dino_head_forward(P, pre, x, exact_zero_rows=False):
    """DINOHead.forward, vision_transformer.py:324-328; weight_norm: w = g * v / ||v||_row (:313).

    exact_zero_rows (NOT the reference's behaviour, default off): every image contributes one all-zero pooled row
    per view (`grid <= length` keeps length+1 rows, dino_vision.py:82-85).  While the head biases are still 0 such a
    row reaches F.normalize as an exact zero vector whose backward multiplies by 1/eps = 1e12; in exact arithmetic
    the incoming gradient there is 0 ((1/K - 1/K)), in fp32 it is rounding residue, so the reference's head-bias
    gradients at those iterations are amplified noise (L2 ~ 1e4 at B=8).  With the flag set, rows whose pre-normalise
    vector is EXACTLY zero carry no gradient - the mathematically exact result, and what a different arithmetic
    (e.g. the HIP path) produces.  Once the biases have moved (after the first update) no row is exactly zero any more
    and the flag changes nothing."""
    x = F.gelu(F.linear(x, P[pre + "mlp.0.weight"], P[pre + "mlp.0.bias"]))
    x = F.gelu(F.linear(x, P[pre + "mlp.2.weight"], P[pre + "mlp.2.bias"]))
    x = F.linear(x, P[pre + "mlp.4.weight"], P[pre + "mlp.4.bias"])
    dead = x.detach().abs().sum(dim=1, keepdim=True) == 0      # rows that reach F.normalize as exact zeros
    x = F.normalize(x, dim=-1, p=2)
    v, g = P[pre + "last_layer.weight_v"], P[pre + "last_layer.weight_g"]
    w = v * (g / v.norm(dim=1, keepdim=True))
    out = F.linear(x, w)
    if exact_zero_rows:
        out = torch.where(dead, out.detach(), out)
    return out


def student_forward(net: Net, images, metrics, target_mask, epoch, drop=None, exact_zero_rows=False):
    """ABIDINOModel.forward with clusters=None, dino_vision.py:51-97."""
    P, spec = net.P, net.spec
    x = torch.cat([images[:, 1], images[:, 2]])
    tokens, taps = backbone_forward(P, "backbone.", x, spec, drop)
    n, T, E = tokens.shape
    region_f = tokens.reshape(n, *spec.grid, E).permute(0, 3, 1, 2)
    seg = seg_head_forward(P, "segmentation.", taps)
    if epoch < 30:
        mask_np = target_mask.detach().cpu().numpy()
    else:
        mask_np = (F.softmax(seg, dim=1)[:, 1] > 0.5).int().detach().cpu().numpy()[: n // 2]
    ids_src = label_batch(mask_np)
    src = torch.from_numpy(ccl_np.idmap_to_planes(ids_src))
    clusters = torch.cat([src, warp_planes(src, metrics)], dim=0)
    vecs, index = region_pool(region_f, clusters)
    rows, new_index = select_rows(vecs, index)
    logits = dino_head_forward(P, "head.", rows, exact_zero_rows)
    return {"instances_view": logits, "mask": seg, "zero": clusters, "index": new_index,
            "idmap": ccl_np.planes_to_idmap(clusters.numpy()), "pool_index": index, "rows": rows,
            "region_f": region_f}


def teacher_forward(net: Net, images, clusters, drop=None):
    """ABIDINOModel.forward with clusters given, dino_vision.py:98-113."""
    P, spec = net.P, net.spec
    x = torch.cat([images[:, 1], images[:, 2]])
    tokens, _ = backbone_forward(P, "backbone.", x, spec, drop)
    n, T, E = tokens.shape
    region_f = tokens.reshape(n, *spec.grid, E).permute(0, 3, 1, 2)
    vecs, index = region_pool(region_f, clusters)
    rows, _ = select_rows(vecs, index)
    return {"instances_view": dino_head_forward(P, "head.", rows), "feature": region_f}


# ------------------------------------------------------------------------------------------------ loss
def teacher_temp_schedule(warmup_temp, temp, warmup_epochs, nepochs):
    """Dino_loss.py:47-51."""
    return np.concatenate((np.linspace(warmup_temp, temp, warmup_epochs),
                           np.ones(nepochs - warmup_epochs) * temp))


def seg_loss(seg_logits, gt):
    """Dino_loss.py:63-66 + SegLoss.cross_entropy 15-26: softmax FIRST, then cross_entropy (double softmax)."""
    p = F.softmax(seg_logits, dim=1).permute(0, 2, 3, 1).reshape(-1, 2)
    return F.cross_entropy(p, gt.reshape(-1).long())


def dino_ce(student_logits, teacher_logits, center, teacher_temp, student_temp=0.1):
    """Dino_loss.py:81-102 with ncrops = 2: the two cross-view pairs, each averaged over its M rows."""
    s1, s2 = (student_logits / student_temp).chunk(2)
    q1, q2 = F.softmax((teacher_logits - center) / teacher_temp, dim=-1).detach().chunk(2)
    l12 = torch.sum(-q1 * F.log_softmax(s2, dim=-1), dim=-1).mean()
    l21 = torch.sum(-q2 * F.log_softmax(s1, dim=-1), dim=-1).mean()
    return (l12 + l21) / 2


def center_update(center, teacher_logits, world_size=1, all_reduce=None, momentum=0.9):
    """Dino_loss.py:133-143: sum over local rows -> all_reduce(SUM) -> / (local_rows * world) -> EMA."""
    bc = teacher_logits.detach().sum(dim=0, keepdim=True)
    if all_reduce is not None:
        all_reduce(bc)
    bc = bc / (teacher_logits.shape[0] * world_size)
    return center * momentum + bc * (1 - momentum)


# ----------------------------------------------------------------------------------------- optimisation
def cosine_iter_schedule(base, final, niter, warmup_iters=0, start_warmup=0.0):
    """modules/utils.py:200-210 (float64 numpy)."""
    warm = np.linspace(start_warmup, base, warmup_iters) if warmup_iters > 0 else np.array([])
    it = np.arange(niter - warmup_iters)
    sched = final + 0.5 * (base - final) * (1 + np.cos(np.pi * it / len(it)))
    sched = np.concatenate((warm, sched))
    assert len(sched) == niter
    return sched


def clip_per_tensor(grads: dict, clip: float):
    """modules/utils.py:132-141: every tensor clipped on ITS OWN L2 norm."""
    norms = {}
    for k, g in grads.items():
        nrm = g.norm(2)
        norms[k] = nrm.item()
        coef = clip / (nrm + 1e-6)
        if coef < 1:
            g.mul_(coef)
    return norms


def param_groups(net: Net):
    """modules/utils.py:643-654: biases and 1-D tensors are not weight-decayed."""
    reg, noreg = [], []
    for k in net.trainable:
        (noreg if (k.endswith(".bias") or net.P[k].dim() == 1) else reg).append(k)
    return reg, noreg


class AdamWState:
    """torch.optim.AdamW defaults (train.py:133): betas (0.9, 0.999), eps 1e-8, per-tensor step counts."""

    def __init__(self):
        self.m, self.v, self.t = {}, {}, {}

    def step(self, net: Net, grads: dict, lr: float, wd: float, b1=0.9, b2=0.999, eps=1e-8):
        reg, _ = param_groups(net)
        reg = set(reg)
        with torch.no_grad():
            for k, g in grads.items():
                p = net.P[k]
                if k not in self.m:
                    self.m[k], self.v[k], self.t[k] = torch.zeros_like(p), torch.zeros_like(p), 0
                self.t[k] += 1
                t = self.t[k]
                p.mul_(1 - lr * (wd if k in reg else 0.0))
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
                p.addcdiv_(self.m[k], denom, value=-lr / (1 - b1 ** t))


def ema_teacher(student: Net, teacher: Net, m: float):
    """train.py:264-272: backbone.* and head.* parameters (buffers and the seg head are not touched)."""
    with torch.no_grad():
        for k, pk in teacher.P.items():
            pk.mul_(m).add_((1 - m) * student.P[k].detach())


def train_iteration(student: Net, teacher: Net, center, opt: AdamWState, batch, epoch, lr, wd, mom,
                    teacher_temp=0.04, clip=3.0, freeze_last_layer=1, world_size=1, all_reduce=None, drop=None,
                    exact_zero_rows=False):
    """train.py:221-272 on CPU.  Returns a record of everything the parity tests look at."""
    images, masks, metrics = batch
    s_out = student_forward(student, images, metrics, masks, epoch, drop, exact_zero_rows)
    t_out = teacher_forward(teacher, images, s_out["zero"])
    masks_image = warp_planes(masks.unsqueeze(1), metrics).squeeze(1)
    gt = torch.cat([masks, masks_image])
    mask_loss = seg_loss(s_out["mask"], gt)
    d_loss = dino_ce(s_out["instances_view"], t_out["instances_view"], center, teacher_temp)
    new_center = center_update(center, t_out["instances_view"], world_size, all_reduce)
    loss = mask_loss + d_loss
    names = [k for k, _ in student.params()]
    gl = torch.autograd.grad(loss, [student.P[k] for k in names], allow_unused=True)
    grads = {k: g.clone() for k, g in zip(names, gl) if g is not None}
    raw = {k: g.clone() for k, g in grads.items()}
    clip_per_tensor(grads, clip)
    clipped = {k: g.clone() for k, g in grads.items()}
    if epoch < freeze_last_layer:          # modules/utils.py:144-149
        grads = {k: g for k, g in grads.items() if "last_layer" not in k}
    opt.step(student, grads, lr, wd)
    ema_teacher(student, teacher, mom)
    return {"loss": loss.item(), "mask_loss": mask_loss.item(), "dino_loss": d_loss.item(), "s_out": s_out,
            "t_out": t_out, "masks_image": masks_image, "center": new_center, "grads_raw": raw,
            "grads_clipped": clipped}
