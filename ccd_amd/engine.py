"""Forward/backward drivers of the HIP kernels (autograd.Function wrappers).

Granularity: one Function per sub-network (backbone, region pooling, DINO head, the two losses), each with a
hand-written backward that launches HIP kernels and writes parameter gradients STRAIGHT into the arena's gradient
buffer (ccd_amd.arena) - autograd only carries activation gradients between the Functions.  Mixed precision:
fp32 residual stream / parameters / statistics, bf16 GEMM and attention operands, fp32 accumulation.

Reference call sites: Dino/modules/vision_transformer.py:107-113, 225-251 (backbone); Dino/model/dino_vision.py:38-49,
80-88 (region pooling + row selection); vision_transformer.py:324-328 (head); Dino/loss/Dino_loss.py:59-143.
"""
from __future__ import annotations

import math

import os

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


class VitSpec:
    def __init__(self, embed_dim, depth, heads, taps, patch=4, eps=1e-6, drop_path_rate=0.0):
        self.E, self.depth, self.heads, self.taps, self.patch, self.eps = embed_dim, depth, heads, tuple(taps), patch, eps
        self.dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]   # vision_transformer.py:150
        self._keep = {}

    def keep_probs(self, device):
        """1 - drop probability of every block, fp32 on `device` (cached)."""
        if device not in self._keep:
            self._keep[device] = torch.tensor([1.0 - d for d in self.dpr], dtype=F32, device=device)
        return self._keep[device]


def _env_switch(key):
    return None if os.environ.get(key) is None else os.environ[key] != "0"


class Fusion:
    """Which fused row-owner kernels the backbone uses.  Read ONCE at import from CCD_FUSE_LN / CCD_FUSE_MLP / CCD_FUSE_LNBWD /
    CCD_SIDE_STREAM (lab switches; None = the measured default for the embedding width); tests and lab scripts may assign the
    attributes.  Nothing on the forward / backward path reads the environment."""
    ln = _env_switch("CCD_FUSE_LN")                # LayerNorm in the epilogue of proj / fc2   (default: E <= 384)
    mlp = _env_switch("CCD_FUSE_MLP")              # fc1 -> GELU -> fc2 -> residual -> LayerNorm in one launch (default: with `ln`)
    lnbwd = _env_switch("CCD_FUSE_LNBWD")          # LayerNorm backward in the epilogue of the data-gradient product (default: on)
    fold_tap = _env_switch("CCD_FOLD_TAP")         # a tap's LayerNorm backward inside the qkv data-gradient product of the block above
    if fold_tap is None:                           # (round 6; CCD_FOLD_TAP=0: the separate ccd_ln_bwd launch per tap)
        fold_tap = True
    store_gact = _env_switch("CCD_STORE_GACT")     # the fused MLP forward also stores gelu(u) for the backward pass (default: off -
    if store_gact is None:                         # a measured tie, 52.9 vs 53.0 ms per step: + 0.47 ms forward, - 0.75 ms gelu'(u)
        store_gact = False                         # product, + 0.2 GB per block of saved activations at B = 256)
    proj_mlp = _env_switch("CCD_FUSE_PROJ")        # proj + residual + LayerNorm-2 in front of the fused MLP, one launch per block half (default: with `mlp`, E <= 384)
    head_loss = _env_switch("CCD_FUSE_HEAD_LOSS")  # last layer + distillation loss in one pass, logits never written (default: on where ccd_head_loss_* take the shape)
    if head_loss is None:
        head_loss = True
    g_bf16 = _env_switch("CCD_G_BF16")             # the backward pass's residual-gradient stream in bf16 (round 6; None = the measured default)
    mlp_bwd = _env_switch("CCD_FUSE_MLP_BWD")      # gelu'(u) product + fc1 data gradient + LayerNorm-2 backward in one launch (ccd_mlp_bwd_fused, round 6;
                                                   # the forward block half then stores gelu(u) too); None = the measured default
    side_stream = bool(_env_switch("CCD_SIDE_STREAM"))
    double_gb = False                       # tests: rotate the two gb buffers of the side-stream mode also without a side stream

    @classmethod
    def resolve(cls, E):
        ln = (cls.ln if cls.ln is not None else E <= 384) and (E <= 384 or E == 512)
        mlp = ln and E % 128 == 0 and (cls.mlp if cls.mlp is not None else True)
        lnbwd = (E <= 384 or E == 512) and E % 8 == 0 and (cls.lnbwd if cls.lnbwd is not None else True)
        return ln, mlp, lnbwd

    @classmethod
    def resolve_g16(cls, E):
        """The gradient of the residual stream as a bf16 tensor between the LayerNorm-backward epilogues (ccd_*_g16): each of the 24
        epilogues of a ViT-Small backward pass reads and rewrites it (9.6 GB per step in fp32).  Row-owner widths only."""
        on = cls.g_bf16 if cls.g_bf16 is not None else G_BF16_DEFAULT
        return bool(on) and cls.resolve(E)[2] and E in (128, 256, 384)

    @classmethod
    def resolve_mlp_bwd(cls, E, H):
        """ccd_mlp_bwd_fused replaces the gelu'(u) product and the LayerNorm-backward product of the MLP branch where the one-launch
        forward block half runs (it hands over gelu(u)) and the gradient stream is bf16."""
        on = cls.mlp_bwd if cls.mlp_bwd is not None else MLP_BWD_DEFAULT
        return bool(on) and cls.resolve_proj(E) and cls.resolve_g16(E) and cls.resolve(E)[2] and E in (256, 384) and H % 128 == 0

    @classmethod
    def resolve_proj(cls, E):
        """ccd_proj_mlp_fused (round 5) replaces ccd_gemm_nt_resid_ln + ccd_mlp_fused where both would run (it cannot store gelu(u))."""
        return cls.resolve(E)[1] and E <= 384 and not cls.store_gact and (cls.proj_mlp if cls.proj_mlp is not None else True)


MLP_BWD_DEFAULT = False       # measured: 46.96 / 47.01 against 46.05 / 46.09 ms per step with it (profiles/r06_mlp_bwd_step_ab.jsonl)
G_BF16_DEFAULT = True         # measured: 46.43 against 46.86 ms per step, every 1e-3 parity gate green (profiles/r06_g_bf16_*)

_DROPPATH_SEED = {"base": None, "calls": 0, "device": None}
DROPPATH_SEED_STRIDE = 0x632BE59BD9B4E019


def set_device_droppath_seed(d_seed):
    """d_seed: int64 [1] on the device (or None).  While set, the DropPath kernel adds *d_seed to its launch-time seed when it
    RUNS: a captured step (pretrain.GraphedTrainingStep) draws new masks at every replay."""
    _DROPPATH_SEED["device"] = d_seed


def _next_droppath_seed():
    """Host-side seed stream: the base is drawn from torch's (seedable) CPU generator at first use."""
    if _DROPPATH_SEED["base"] is None:
        _DROPPATH_SEED["base"] = int(torch.randint(0, 2 ** 62, (1,)).item())
    _DROPPATH_SEED["calls"] += 1
    return _DROPPATH_SEED["base"] + _DROPPATH_SEED["calls"] * DROPPATH_SEED_STRIDE


# ------------------------------------------------------------------------------------------------ pos-embed resampling
def bicubic_resample_matrix(n_src_side=16, out_h=8, out_w=32):
    """The linear map of interpolate_pos_encoding (vision_transformer.py:182-201) as a dense [out_h*out_w, n*n]
    fp32 matrix: F.interpolate(mode='bicubic', align_corners=False) with the PASSED scale factors
    ((out+0.1)/n) driving the coordinate transform, cubic coefficient A = -0.75, border-clamped taps.
    Restated from ATen's upsample_bicubic2d (area_pixel_compute_source_index(scale, dst, False, cubic=True),
    get_cubic_upsample_coefficients)."""
    import numpy as np

    def coeffs(t):
        a = np.float32(-0.75)
        t = np.float32(t)
        c1 = lambda x: ((a + np.float32(2)) * x - (a + np.float32(3))) * x * x + np.float32(1)
        c2 = lambda x: ((a * x - np.float32(5) * a) * x + np.float32(8) * a) * x - np.float32(4) * a
        return [c2(t + np.float32(1)), c1(t), c1(np.float32(1) - t), c2(np.float32(2) - t)]

    def axis(n_in, n_out):
        sf = (n_out + 0.1) / math.sqrt(n_in * n_in)            # the scale_factor the reference passes
        scale = np.float32(1.0 / sf)                           # compute_scales_value: 1 / scale_factor
        taps = []
        for o in range(n_out):
            src = scale * (np.float32(o) + np.float32(0.5)) - np.float32(0.5)
            fl = np.floor(src)
            w = coeffs(src - fl)
            idx = [min(max(int(fl) - 1 + k, 0), n_in - 1) for k in range(4)]
            taps.append((idx, w))
        return taps

    import numpy as np
    ty, tx = axis(n_src_side, out_h), axis(n_src_side, out_w)
    m = np.zeros((out_h * out_w, n_src_side * n_src_side), dtype=np.float32)
    for oy in range(out_h):
        for ox in range(out_w):
            for iy, wy in zip(*ty[oy]):
                for ix, wx in zip(*tx[ox]):
                    m[oy * out_w + ox, iy * n_src_side + ix] += np.float32(wy) * np.float32(wx)
    return torch.from_numpy(m)


# ------------------------------------------------------------------------------------------------------ backbone
class _BlockCtx:
    __slots__ = ("x_in", "y1", "mean1", "rstd1", "qkv", "att", "lse", "x_mid", "y2", "mean2", "rstd2", "u", "gact",
                 "ds1", "ds2")


def backbone_forward(arena, pre, spec: VitSpec, img, resample, save, training, need_taps=True):
    """img [N,3,32,128] fp32 -> (tokens bf16 [N*256,E], [tap bf16 [N*256,E]] * len(taps), ctx or None)."""
    E, N = spec.E, img.shape[0]
    R = N * 256
    dev = img.device
    pos = ops.small_matmul(resample, arena.w(pre + "pos_embed").view(-1, E), torch.empty((256, E), dtype=F32, device=dev))
    x = ops.patch_embed_fwd(img, arena.w(pre + "patch_embed.proj.weight"), arena.w(pre + "patch_embed.proj.bias"), pos)
    ctxs, taps, tap_ctx = [], [], []
    scale = (E // spec.heads) ** -0.5
    # LayerNorm folded into the epilogue of the residual product that finishes its input rows (full-row kernel,
    # E <= 384 or 512): proj -> norm2 of the same block, fc2 -> norm1 of the next block / the final norm
    # (E = 512 - vit_base: the row-owner kernels exist (`CCD_FUSE_LN=1` forces them) but lose there - 256 accumulator registers
    # per lane leave a 3-slot weight ring and spills: 51.9 ms per step against 47.0 with only the LayerNorm-backward product
    # fused, 47.3 unfused; B = 128, one MI355X)
    fuse_ln, fuse_mlp, _ = Fusion.resolve(E)
    fuse_proj = Fusion.resolve_proj(E)
    # (the one-launch MLP backward wants gelu(u) from the forward kernel: ccd_proj_mlp_fused_gact)
    bwd_fused = fuse_proj and Fusion.resolve_mlp_bwd(E, arena.w(f"{pre}blocks.0.mlp.fc1.weight").shape[0])
    # the whole MLP branch in one kernel (csrc/kernels/mlp_fused.h): the hidden activation never reaches HBM; when
    # activations are saved only the bf16 pre-activation u is stored and backward re-derives gelu(u) in the epilogue
    # that already reads u (ccd_gemm_nt, EPI_DGELU with a second output)
    pending = None                               # (y, mean, rstd) of the coming norm1, made by the previous fc2
    # DropPath: per-(block, branch, sample) keep mask / keep_prob (vision_transformer.py:27-35), one kernel per pass
    scales = ops.droppath_scales(spec.keep_probs(dev), N, _next_droppath_seed(), _DROPPATH_SEED["device"]) if training and max(spec.dpr) > 0.0 else None
    for i in range(spec.depth):
        b = f"{pre}blocks.{i}."
        c = _BlockCtx()
        c.ds1 = c.ds2 = None
        if scales is not None and spec.dpr[i] > 0.0:
            c.ds1, c.ds2 = scales[i, 0], scales[i, 1]
        c.x_in = x
        if pending is None:
            c.y1, c.mean1, c.rstd1 = ops.ln_fwd(x, arena.w(b + "norm1.weight"), arena.w(b + "norm1.bias"), spec.eps)
        else:
            c.y1, c.mean1, c.rstd1 = pending
        c.qkv = ops.gemm_nt(c.y1, arena.wb(b + "attn.qkv.weight"), bias=arena.w(b + "attn.qkv.bias"))
        c.att, c.lse = ops.attention_fwd(c.qkv.view(N, 256, 3 * E), spec.heads, scale)
        # (a dropped MLP branch reads x_mid back, so the one-launch block half needs x_mid written: a pass that keeps nothing AND
        # draws DropPath masks - train() under no_grad - takes the two-launch path for that block)
        proj_here = fuse_proj and (save or c.ds2 is None)
        if proj_here:
            # the block's second half in ONE launch: x_mid and y2 stay in registers (a pass that saves nothing writes neither)
            nxt = f"{pre}blocks.{i + 1}.norm1." if i + 1 < spec.depth else pre + "norm."
            # a segmentation tap behind this block is one more LayerNorm of the same rows: the kernel emits it (same statistics)
            tap_j = len(taps) if (need_taps and i + 1 in spec.taps) else None
            tap_kw = {} if tap_j is None else dict(tap_gamma=arena.w(f"{pre}norm_seg.{tap_j}.weight"), tap_beta=arena.w(f"{pre}norm_seg.{tap_j}.bias"))
            x, y_n, mean_n, rstd_n, kept, *tap_out = ops.proj_mlp_fused(
                c.att.view(R, E), arena.wb(b + "attn.proj.weight"), arena.w(b + "attn.proj.bias"), resid=x, rowscale1=c.ds1,
                gamma2=arena.w(b + "norm2.weight"), beta2=arena.w(b + "norm2.bias"), w1=arena.wb(b + "mlp.fc1.weight"),
                b1=arena.w(b + "mlp.fc1.bias"), w2=arena.wb(b + "mlp.fc2.weight"), b2=arena.w(b + "mlp.fc2.bias"), rowscale2=c.ds2,
                rows_per_sample=256, gamma=arena.w(nxt + "weight"), beta=arena.w(nxt + "bias"), eps=spec.eps, save=save,
                store_gact=bool(save and bwd_fused), **tap_kw)
            c.gact = None
            if save:
                c.x_mid, c.y2, c.mean2, c.rstd2, c.u, *g_kept = kept
                c.gact = g_kept[0] if g_kept else None
            pending = [y_n, mean_n, rstd_n]
        elif fuse_ln:
            c.x_mid, c.y2, c.mean2, c.rstd2 = ops.gemm_nt_resid_ln(
                c.att.view(R, E), arena.wb(b + "attn.proj.weight"), bias=arena.w(b + "attn.proj.bias"), resid=x,
                rowscale=c.ds1, rows_per_sample=256, gamma=arena.w(b + "norm2.weight"), beta=arena.w(b + "norm2.bias"),
                eps=spec.eps)
        else:
            c.x_mid = ops.gemm_nt(c.att.view(R, E), arena.wb(b + "attn.proj.weight"), epilogue=ops.EPI_RESID,
                                  bias=arena.w(b + "attn.proj.bias"), resid=x, rowscale=c.ds1, rows_per_sample=256)
            c.y2, c.mean2, c.rstd2 = ops.ln_fwd(c.x_mid, arena.w(b + "norm2.weight"), arena.w(b + "norm2.bias"), spec.eps)
        if proj_here:
            pass
        elif fuse_mlp:
            nxt = f"{pre}blocks.{i + 1}.norm1." if i + 1 < spec.depth else pre + "norm."
            # (store_gact: gelu(u) leaves the forward kernel too - its packed second-product operands ARE that tensor - so the
            # backward's gelu'(u) product gathers one table instead of two and writes du only)
            keep_g = bool(save and Fusion.store_gact)
            res = ops.mlp_fused(
                c.y2, arena.wb(b + "mlp.fc1.weight"), arena.w(b + "mlp.fc1.bias"), arena.wb(b + "mlp.fc2.weight"),
                arena.w(b + "mlp.fc2.bias"), resid=c.x_mid, rowscale=c.ds2, rows_per_sample=256,
                gamma=arena.w(nxt + "weight"), beta=arena.w(nxt + "bias"), eps=spec.eps, store_u=save, store_gact=keep_g)
            x, y_n, mean_n, rstd_n, c.u = res[:5]
            c.gact = res[5] if keep_g else None
            pending = [y_n, mean_n, rstd_n]
        elif fuse_ln:
            c.u, c.gact = ops.gemm_nt(c.y2, arena.wb(b + "mlp.fc1.weight"), epilogue=ops.EPI_GELU,
                                      bias=arena.w(b + "mlp.fc1.bias"), store_u=save)   # u only feeds gelu' in backward
            nxt = f"{pre}blocks.{i + 1}.norm1." if i + 1 < spec.depth else pre + "norm."
            x, *pending = ops.gemm_nt_resid_ln(
                c.gact, arena.wb(b + "mlp.fc2.weight"), bias=arena.w(b + "mlp.fc2.bias"), resid=c.x_mid, rowscale=c.ds2,
                rows_per_sample=256, gamma=arena.w(nxt + "weight"), beta=arena.w(nxt + "bias"), eps=spec.eps)
        else:
            c.u, c.gact = ops.gemm_nt(c.y2, arena.wb(b + "mlp.fc1.weight"), epilogue=ops.EPI_GELU,
                                      bias=arena.w(b + "mlp.fc1.bias"), store_u=save)
            x = ops.gemm_nt(c.gact, arena.wb(b + "mlp.fc2.weight"), epilogue=ops.EPI_RESID,
                            bias=arena.w(b + "mlp.fc2.bias"), resid=c.x_mid, rowscale=c.ds2, rows_per_sample=256)
        if not save:
            c.y1 = c.qkv = c.att = c.y2 = c.u = c.gact = None
        ctxs.append(c if save else None)
        if need_taps and i + 1 in spec.taps:
            j = len(taps)
            if proj_here:
                t, mu, rs = tap_out[0], mean_n, rstd_n           # (LayerNorm statistics depend on the rows only)
            else:
                t, mu, rs = ops.ln_fwd(x, arena.w(f"{pre}norm_seg.{j}.weight"), arena.w(f"{pre}norm_seg.{j}.bias"), spec.eps)
            taps.append(t)
            tap_ctx.append((i, x, mu, rs))
    if pending is None:
        tokens, mu, rs = ops.ln_fwd(x, arena.w(pre + "norm.weight"), arena.w(pre + "norm.bias"), spec.eps)
    else:
        tokens, mu, rs = pending
    ctx = (ctxs, tap_ctx, (x, mu, rs), img) if save else None
    return tokens, taps, ctx


class _SideStream:
    """Weight-gradient products (dW = dY^T X, TN GEMMs) and bias column sums have no consumer inside the backward
    pass, so they run on a second HIP stream next to the data-gradient chain: while the main stream sits in an
    HBM-bound LayerNorm backward, the side stream's MFMA-bound GEMM fills the matrix pipes (and vice versa for the
    column sums).  Ordering is by events; tensors handed to the side stream are `record_stream`ed so the caching
    allocator cannot recycle them early.  MEASURED (B=256, MI355X): no gain - 69.25 vs 69.09 ms/step - the persistent
    GEMMs already hold every workgroup slot of every CU, so kernels of the other stream only get in as slots free up.
    Kept as an opt-in (CCD_SIDE_STREAM=1) for later kernels that leave room; off by default and off the GPU."""

    _streams = {}

    def __init__(self, device):
        self.on = device.type == "cuda" and Fusion.side_stream
        if self.on:
            key = device.index if device.index is not None else torch.cuda.current_device()
            if key not in _SideStream._streams:
                # high priority: when both streams have workgroups waiting, the 144-KiB weight-gradient workgroups go first and
                # the small LayerNorm-backward blocks fill in beside them (the other order leaves no room for the big ones)
                _SideStream._streams[key] = torch.cuda.Stream(device=device, priority=-1)
            self.side = _SideStream._streams[key]
            self.main = torch.cuda.current_stream(device)
        self.pending = []                      # events of side work since the last join

    def run(self, fn, *tensors):
        """fn() is enqueued on the side stream after everything already enqueued on the main stream."""
        if not self.on:
            fn()
            return None
        ready = torch.cuda.Event()
        ready.record(self.main)
        self.side.wait_event(ready)
        with torch.cuda.stream(self.side):
            fn()
        for t in tensors:
            t.record_stream(self.side)
        done = torch.cuda.Event()
        done.record(self.side)
        self.pending.append(done)
        return done

    def wait(self, event):
        if self.on and event is not None:
            self.main.wait_event(event)

    def join(self):
        """Main stream waits for all side work issued so far (gradients of a block are final after this)."""
        if self.on:
            for ev in self.pending:
                self.main.wait_event(ev)
        self.pending = []


def backbone_backward(arena, pre, spec: VitSpec, ctx, d_tokens, d_taps, resample, on_block_done=None):
    """Consumes bf16 gradients of the final-norm tokens and the taps; fills the arena gradient slots of `pre`*.

    g is the fp32 gradient of the residual stream.  Every LayerNorm backward that is the LAST writer of g before a
    residual branch also emits gb = bf16(g * DropPath scale of that branch) and the branch's output-bias gradient
    (column sums of gb) in the same pass (layernorm.h), so no separate cast / column-sum kernels run per branch."""
    ctxs, tap_ctx, (x_last, mu, rs), img = ctx
    E, N = spec.E, img.shape[0]
    R = N * 256
    dev = img.device
    g = torch.empty((R, E), dtype=BF16 if Fusion.resolve_g16(E) else F32, device=dev)
    tap_at = {i: (j, x, m, r) for j, (i, x, m, r) in enumerate(tap_ctx) if d_taps[j] is not None}
    scale = (E // spec.heads) ** -0.5
    top = spec.depth - 1
    # LayerNorm backward folded into the epilogue of the data-gradient product in front of it (ccd_gemm_nt_lnbwd, N <= 384 or 512)
    fuse_lnbwd = Fusion.resolve(E)[2]
    side = _SideStream(dev)
    grad_fresh = getattr(arena, "grad_fresh", False)     # the gradient arena was zeroed since the last backward pass
    arena.grad_fresh = False
    # gb = bf16(g * DropPath scale), the gradient that enters a residual branch.  With the side stream on it lives in TWO
    # buffers used in turn: a LayerNorm backward writes the next branch's gb while the weight-gradient launch of the
    # previous branch (side stream) still reads the old one.
    # (the one-launch MLP backward reads the branch's gb and writes the next branch's: two buffers as well)
    bwd_fused = fuse_lnbwd and ctxs[top] is not None and ctxs[top].gact is not None and ctxs[top].u is not None and \
        Fusion.resolve_mlp_bwd(E, ctxs[top].u.shape[1]) and ops.mlp_bwd_fused_supported(g, E, ctxs[top].u.shape[1])
    gbuf = [torch.empty((R, E), dtype=BF16, device=dev) for _ in range(2 if (side.on or Fusion.double_gb or bwd_fused) else 1)]
    readers = [None] * len(gbuf)      # event of the last side-stream launch that reads each buffer
    cur = [0]

    def gb_read():
        return gbuf[cur[0]]

    def gb_write():                   # the buffer the next writer fills (it becomes the current one)
        cur[0] = (cur[0] + 1) % len(gbuf)
        side.wait(readers[cur[0]])
        readers[cur[0]] = None
        return gbuf[cur[0]]

    def mlp_tail(i):            # what the MLP branch of block i wants from the writer in front of it
        return dict(gb=gb_write(), rowscale=ctxs[i].ds2, rows_per_sample=256, dbias=arena.g(f"{pre}blocks.{i}.mlp.fc2.bias"))

    have_gb = False
    if d_tokens is not None:
        tail = mlp_tail(top) if top not in tap_at else {}
        ops.ln_bwd(d_tokens.reshape(R, E), x_last, mu, rs, arena.w(pre + "norm.weight"), g, arena.g(pre + "norm.weight"),
                   arena.g(pre + "norm.bias"), accumulate=False, **tail)
        have_gb = bool(tail)
    else:
        g.zero_()
    # A tap behind block i is a second LayerNorm of the rows that norm1 of block i + 1 normalises (same statistics): its backward pass
    # rides in the epilogue of block i + 1's qkv data-gradient product (ccd_gemm_nt_lnbwd_tap_g16) instead of a launch of its own
    fold_taps = fuse_lnbwd and Fusion.fold_tap and ops.lnbwd_tap_supported(g, E)
    for i in reversed(range(spec.depth)):
        b = f"{pre}blocks.{i}."
        c = ctxs[i]
        if i in tap_at and not (fold_taps and i + 1 < spec.depth):
            j, xt, m, r = tap_at[i]
            ops.ln_bwd(d_taps[j].reshape(R, E), xt, m, r, arena.w(f"{pre}norm_seg.{j}.weight"), g,
                       arena.g(f"{pre}norm_seg.{j}.weight"), arena.g(f"{pre}norm_seg.{j}.bias"), accumulate=True,
                       **mlp_tail(i))
            have_gb = True
        if not have_gb:          # only when no gradient reached the final norm: plain cast + column sum
            gbw = gb_write()
            ops.scale_cast_rows(g if g.dtype == F32 else g.float(), gbw, c.ds2, 256)
            ops.colsum_bf16(gbw, arena.g(b + "mlp.fc2.bias"))
        # ---- MLP branch: x_out = x_mid + ds2 * fc2(gelu(fc1(LN2(x_mid))))
        gact, y2, att, y1 = c.gact, c.y2, c.att, c.y1
        gb = gb_read()
        # (The whole chain - gelu'(u) product, fc1 data gradient, LayerNorm-2 backward - as ONE row-owner kernel, the mirror of mlp_fused.h,
        # was built twice.  Round 3: 836 us per block against 717 for the two launches (u streamed in row-per-lane through the weight
        # ring's vmcnt queue).  Round 6 (mlp_bwd.h: u by LDS-DMA two chunks ahead, gelu(u) handed over by the forward kernel): 0.558
        # against 0.598 ms in the lab, 0.607 against 0.605 in the step - and the forward block half pays 0.088 ms per block for storing
        # gelu(u): 46.96 / 47.01 against 46.05 / 46.09 ms per step.  Kept behind CCD_FUSE_MLP_BWD=1, off by default.
        # profiles/r03_mlp_bwd_fused_ab.jsonl, profiles/r06_mlp_bwd_lab.jsonl, profiles/r06_mlp_bwd_step_ab.jsonl.)
        fused_here = bwd_fused and gact is not None
        if fused_here:
            # gelu'(u) product, fc1 data gradient and LayerNorm-2 backward in ONE launch (mlp_bwd.h): du is written once, for the weight
            # gradients, and never read back; the next branch's gb goes to the other buffer (the weight gradients still read this one)
            old = cur[0]
            du = ops.mlp_bwd_fused(gb, arena.wbt(b + "mlp.fc2.weight"), arena.wbt(b + "mlp.fc1.weight"), c.u,
                                   db1=arena.g(b + "mlp.fc1.bias"), x=c.x_mid, mean=c.mean2, rstd=c.rstd2, gamma=arena.w(b + "norm2.weight"),
                                   g=g, dgamma=arena.g(b + "norm2.weight"), dbeta=arena.g(b + "norm2.bias"), gb_out=gb_write(),
                                   rowscale=c.ds1, rows_per_sample=256, dbias=arena.g(b + "attn.proj.bias"), accumulate=True)
        elif gact is None:        # fused-MLP forward kept only u: gelu(u) comes out of the gelu'(u) epilogue below
            gact = torch.empty_like(c.u)
            du = ops.gemm_nt(gb, arena.wbt(b + "mlp.fc2.weight"), epilogue=ops.EPI_DGELU, aux=c.u, out2=gact,
                             colsum=arena.g(b + "mlp.fc1.bias"))
        else:
            du = ops.gemm_nt(gb, arena.wbt(b + "mlp.fc2.weight"), epilogue=ops.EPI_DGELU, aux=c.u,
                             colsum=arena.g(b + "mlp.fc1.bias"))

        # both weight gradients of the MLP in ONE launch (ccd_gemm_tn_pair: the same rows, one atomic epilogue per workgroup)
        def mlp_grads(gb=gb, du=du, gact=gact):
            ops.gemm_tn_pair(gb, gact, arena.g(b + "mlp.fc2.weight"), du, y2, arena.g(b + "mlp.fc1.weight"))
        if fused_here:
            readers[old] = side.run(mlp_grads, gb, gact, du, y2)
        elif fuse_lnbwd:       # dy2 = du . W1 never leaves the chip: LayerNorm-2's backward is the product's epilogue
            readers[cur[0]] = side.run(mlp_grads, gb, gact, du, y2)
            ops.gemm_nt_lnbwd(du, arena.wbt(b + "mlp.fc1.weight"), c.x_mid, c.mean2, c.rstd2, arena.w(b + "norm2.weight"), g,
                              arena.g(b + "norm2.weight"), arena.g(b + "norm2.bias"), accumulate=True, gb=gb_write(),
                              rowscale=c.ds1, rows_per_sample=256, dbias=arena.g(b + "attn.proj.bias"))
        else:
            # unfused: the product first, THEN the weight-gradient launch (side stream: it starts once the product is done)
            # and the HBM-bound LayerNorm backward beside it - ln_bwd_kernel's 6 KiB of LDS and 44 registers fit next to a
            # 144-KiB gemm_tn384 workgroup.  MEASURED (round 3, profiles/r03_side_stream_ab.jsonl): the two do run
            # concurrently, and each takes as much longer as the other lasts - no gain; the fused path is the default.
            dy2 = ops.gemm_nt(du, arena.wbt(b + "mlp.fc1.weight"))
            readers[cur[0]] = side.run(mlp_grads, gb, gact, du, y2)
            ops.ln_bwd(dy2, c.x_mid, c.mean2, c.rstd2, arena.w(b + "norm2.weight"), g, arena.g(b + "norm2.weight"),
                       arena.g(b + "norm2.bias"), accumulate=True, gb=gb_write(), rowscale=c.ds1, rows_per_sample=256,
                       dbias=arena.g(b + "attn.proj.bias"))
        del du
        # ---- attention branch: x_mid = x_in + ds1 * proj(attn(qkv(LN1(x_in))))
        gb = gb_read()
        d_att = ops.gemm_nt(gb, arena.wbt(b + "attn.proj.weight"))
        # the qkv-bias gradient = column sums of d_qkv without a pass over d_qkv (round 2: 0.95 ms per step of colsum launches):
        # q part from the dQ kernel's fp32 tiles, k part identically zero, v part = colsum(d_att) = colsum(gb) . Wproj, and
        # colsum(gb) is proj.bias's gradient, final since the LayerNorm-2 backward above (ops.attention_bwd, include/ccd_hip.h)
        if grad_fresh:
            dcs, dcs_mat = arena.g(b + "attn.proj.bias"), arena.w(b + "attn.proj.weight")
        else:       # gradients are being accumulated over several backward passes: the slot holds more than this pass's sum
            dcs, dcs_mat = ops.colsum_bf16(d_att, torch.zeros(E, dtype=F32, device=dev)), None
        d_qkv = ops.attention_bwd(c.qkv.view(N, 256, 3 * E), c.att, d_att.view(N, 256, E), c.lse, spec.heads, scale,
                                  d_bias=arena.g(b + "attn.qkv.bias"), dout_colsum=dcs, dout_colsum_mat=dcs_mat)
        d_qkv = d_qkv.view(R, 3 * E)

        def qkv_grads(gb=gb, d_qkv=d_qkv):
            # proj.weight's gradient waits for qkv.weight's: one launch for both
            ops.gemm_tn_pair(gb, att.view(R, E), arena.g(b + "attn.proj.weight"), d_qkv, y1, arena.g(b + "attn.qkv.weight"))
        fold_here = fold_taps and (i - 1) in tap_at
        want_tail = i > 0 and ((i - 1) not in tap_at or fold_here)
        if fuse_lnbwd:
            readers[cur[0]] = side.run(qkv_grads, gb, att, d_qkv, y1)
            tail = mlp_tail(i - 1) if want_tail else {}
            if fold_here:      # the tap behind block i - 1: x_in of this block IS the tap's input, mean1 / rstd1 its statistics
                j = tap_at[i - 1][0]
                tail["tap"] = (d_taps[j].reshape(R, E), arena.w(f"{pre}norm_seg.{j}.weight"), arena.g(f"{pre}norm_seg.{j}.weight"),
                               arena.g(f"{pre}norm_seg.{j}.bias"))
            ops.gemm_nt_lnbwd(d_qkv, arena.wbt(b + "attn.qkv.weight"), c.x_in, c.mean1, c.rstd1, arena.w(b + "norm1.weight"), g,
                              arena.g(b + "norm1.weight"), arena.g(b + "norm1.bias"), accumulate=True, **tail)
        else:
            dy1 = ops.gemm_nt(d_qkv, arena.wbt(b + "attn.qkv.weight"))
            readers[cur[0]] = side.run(qkv_grads, gb, att, d_qkv, y1)
            tail = mlp_tail(i - 1) if want_tail else {}
            ops.ln_bwd(dy1, c.x_in, c.mean1, c.rstd1, arena.w(b + "norm1.weight"), g, arena.g(b + "norm1.weight"),
                       arena.g(b + "norm1.bias"), accumulate=True, **tail)
        have_gb = bool(tail)
        ctxs[i] = None
        if on_block_done is not None:
            side.join()                                      # the block's weight gradients are final for the reducer
            on_block_done(b)
    side.join()
    d_pos_rs = torch.zeros((256, E), dtype=F32, device=dev)
    ops.patch_embed_bwd(img, g, arena.g(pre + "patch_embed.proj.weight").view(E, -1),
                        arena.g(pre + "patch_embed.proj.bias"), d_pos_rs)
    ops.small_matmul(resample, d_pos_rs, arena.g(pre + "pos_embed").view(-1, E), trans_a=True, accumulate=True)
    if on_block_done is not None:
        on_block_done(pre + "patch_embed.")


class ForwardHop:
    """Where BackboneFn.forward runs while set as engine.FORWARD_HOP: `stream` (CU-masked), persistent grids sized to leave
    `reserve` compute units to the other partition; join=False leaves it to the caller to make its stream wait for `stream`."""

    def __init__(self, stream, reserve, join=True):
        self.stream, self.reserve, self.join = stream, int(reserve), bool(join)


FORWARD_HOP = None


class BackboneFn(torch.autograd.Function):
    """tokens, tap0, tap1, tap2 = BackboneFn.apply(anchor, img, module)   (anchor: any tensor that requires grad)."""

    @staticmethod
    def forward(ctx, anchor, img, module, need_taps=True):
        save = ctx.needs_input_grad[0]      # False under no_grad / frozen teacher
        hop = FORWARD_HOP
        if hop is not None and img.is_cuda:
            # the pass runs on a CU-masked stream (ccd_amd/streams.py) beside the other network's; the stream switch happens INSIDE
            # the Function, so autograd still files the node under the caller's stream and the backward pass gets the whole chip
            main = torch.cuda.current_stream(img.device)
            hop.stream.wait_stream(main)
            # img (and the module's resample matrix) were allocated on the caller's stream: tell the caching allocator that the hop
            # stream reads them, or a join=False pass could see its input block handed to the caller's next allocation
            img.record_stream(hop.stream)
            if module.resample.is_cuda:
                module.resample.record_stream(hop.stream)
            with torch.cuda.stream(hop.stream), ops.policy(cu_reserve=hop.reserve, cu_reserve_window=-1):
                tokens, taps, saved = backbone_forward(module.arena, module.arena_prefix, module.spec, img, module.resample,
                                                       save, module.training, need_taps)
            if hop.join:
                main.wait_stream(hop.stream)
        else:
            tokens, taps, saved = backbone_forward(module.arena, module.arena_prefix, module.spec, img, module.resample,
                                                   save, module.training, need_taps)
        ctx.module, ctx.saved = module, saved
        N, E = img.shape[0], module.spec.E
        outs = [tokens.view(N, 256, E)] + [t.view(N, 256, E) for t in taps]
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_tokens, *d_taps):
        m = ctx.module
        cast = lambda t: None if t is None else t.contiguous().to(BF16)
        backbone_backward(m.arena, m.arena_prefix, m.spec, ctx.saved, cast(d_tokens), [cast(t) for t in d_taps],
                          m.resample, m.grad_ready_hook)
        ctx.saved = None
        return None, None, None, None


# ---------------------------------------------------------------------------------------------- region pooling
class Selection:
    """Device-resident result of the character-region bookkeeping for one batch (both views)."""

    def __init__(self, idmap, batch):
        self.idmap, self.batch = idmap, batch                      # uint8 [2B,32,128]
        self.tok_plane, self.tok_coef, self.present = ops.region_stats(idmap)
        self.nsel, self.offset, self.total, self.new_index = ops.select_scan(self.present, batch)
        self.max_rows = 2 * 26 * batch
        self._m = None

    @property
    def M(self):                                                   # host sync; only for API-compat accessors / logging
        if self._m is None:
            self._m = int(self.total.item())
        return self._m

    def dense(self):
        """The reference's `clusters` tensor [2B,26,32,128] fp32 (dino_vision.py:78)."""
        return ops.idmap_to_planes(self.idmap)


class RegionPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, sel: Selection):
        E = tokens.shape[-1]
        rows = torch.zeros((sel.max_rows, E), dtype=BF16, device=tokens.device)
        ops.region_pool_fwd(tokens.contiguous(), sel.tok_plane, sel.tok_coef, sel.nsel, sel.offset, sel.total, rows,
                            sel.batch)
        ctx.sel, ctx.shape = sel, tokens.shape
        return rows

    @staticmethod
    def backward(ctx, d_rows):
        sel = ctx.sel
        d_feat = torch.empty(ctx.shape, dtype=BF16, device=d_rows.device)
        ops.region_pool_bwd(d_rows.contiguous().to(BF16), sel.tok_plane, sel.tok_coef, sel.nsel, sel.offset, sel.total,
                            d_feat, sel.batch)
        return d_feat, None


# ------------------------------------------------------------------------------------------------------- DINO head
class LazyLogits:
    """logits = zn @ w^T of a DINO head that nobody has asked for yet (round 6).  The distillation loss takes the two factors
    (ccd_head_loss_fwd / _bwd: a logit lives in a register for as long as it takes to fold it into its row's softmax state); whoever
    wants the [max_rows, K] fp32 tensor itself calls tensor() and pays for the product then.  `stub` is the 1-element autograd
    stand-in HeadFn returned: the bf16 logit gradient travels to HeadFn.backward through _BF16_LOGIT_GRADS under its address."""

    def __init__(self, zn, w, d_total, rows_mul):
        self.zn, self.w, self.d_total, self.rows_mul = zn, w, d_total, rows_mul
        self.stub = None
        self._tensor = None

    @property
    def shape(self):
        return torch.Size((self.zn.shape[0], self.w.shape[0]))

    @property
    def device(self):
        return self.zn.device

    def detach(self):
        return self

    def tensor(self):
        """The [max_rows, K] fp32 matrix itself (rows < rows_mul * M computed), outside autograd: for inspection / API parity."""
        if self._tensor is None:
            self._tensor = torch.empty(tuple(self.shape), dtype=F32, device=self.zn.device)
            ops.gemm_nt(self.zn, self.w, epilogue=ops.EPI_F32, out=self._tensor, m_fastest=1, d_rows=self.d_total, rows_mul=self.rows_mul)
        return self._tensor


_LAZY_LOGITS = {}        # stub address -> LazyLogits, between HeadFn.forward and the caller of HeadFn.apply


def logits_tensor(x):
    return x.tensor() if isinstance(x, LazyLogits) else x


def head_forward(arena, pre, rows, d_total, save, rows_mul=2, lazy=False):
    """rows bf16 [max_rows, E] -> logits fp32 [max_rows, K] (only the first rows_mul*d_total[0] rows are computed); with `lazy`,
    where the fused head + loss kernels take the shape, a LazyLogits instead (the product is left to the loss)."""
    dev = rows.device
    dyn = dict(d_rows=d_total, rows_mul=rows_mul)
    u0, a0 = ops.gemm_nt(rows, arena.wb(pre + "mlp.0.weight"), epilogue=ops.EPI_GELU, bias=arena.w(pre + "mlp.0.bias"), **dyn)
    u1, a1 = ops.gemm_nt(a0, arena.wb(pre + "mlp.2.weight"), epilogue=ops.EPI_GELU, bias=arena.w(pre + "mlp.2.bias"), **dyn)
    z = ops.gemm_nt(a1, arena.wb(pre + "mlp.4.weight"), bias=arena.w(pre + "mlp.4.bias"), **dyn)
    zn = torch.empty_like(z)
    inv = torch.empty(z.shape[0], dtype=F32, device=dev)
    ops.l2norm_fwd(z, zn, inv, d_rows=d_total, rows_mul=rows_mul)
    v, gw = arena.w(pre + "last_layer.weight_v"), arena.w(pre + "last_layer.weight_g")
    K, D = v.shape
    w = torch.empty((K, D), dtype=BF16, device=dev)
    w_t = torch.empty((D, K), dtype=BF16, device=dev) if save else None
    winv = torch.empty(K, dtype=F32, device=dev)
    ops.weightnorm_fwd(v, gw, w, w_t, winv)
    saved = (rows, u0, a0, u1, a1, z, zn, inv, w_t, winv) if save else None
    if lazy and rows_mul == 2 and Fusion.head_loss and ops.head_loss_supported(K, D, rows.shape[0]):
        return LazyLogits(zn, w, d_total, rows_mul), saved
    logits = torch.empty((rows.shape[0], K), dtype=F32, device=dev)
    ops.gemm_nt(zn, w, epilogue=ops.EPI_F32, out=logits, m_fastest=1, **dyn)
    if not save:
        # the factors of a pass that keeps nothing (the teacher): DINOLoss.update_center takes the logits' column sums from them - a
        # matrix-vector product over w instead of a pass over the [rows, K] logits.  ONE slot: whatever an earlier pass left is dropped.
        _LOGIT_FACTORS.clear()
        if D % 256 == 0:
            _LOGIT_FACTORS[logits.data_ptr()] = (zn, w)
    return logits, saved


_LOGIT_FACTORS = {}      # logits buffer address -> (zn bf16 [rows, D], w bf16 [K, D]) with logits = zn @ w^T (see head_forward)


def logit_column_sums(logits, d_total, out, rows_mul=2):
    """out[k] += sum over the first rows_mul * d_total[0] rows of logits[:, k] - through the factors head_forward parked for this
    buffer when it has them (colsum of zn, then w . that), by a pass over the logits otherwise."""
    if isinstance(logits, LazyLogits):
        fac = (logits.zn, logits.w) if logits.w.shape[1] % 256 == 0 else None
        if fac is None:
            logits = logits.tensor()
    else:
        fac = _LOGIT_FACTORS.pop(logits.data_ptr(), None)
    if fac is None or fac[0].shape[0] != logits.shape[0]:
        ops.colsum_f32(logits, out, d_rows=d_total, rows_mul=rows_mul)
        return out
    zn, w = fac
    zsum = torch.zeros(zn.shape[1], dtype=F32, device=zn.device)
    ops.colsum_bf16(zn, zsum, d_rows=d_total, rows_mul=rows_mul)
    return ops.matvec_bf16(w, zsum, out)


def head_backward(arena, pre, saved, d_logits, d_total, last_layer_trainable_g, rows_mul=2):
    rows, u0, a0, u1, a1, z, zn, inv, w_t, winv = saved
    dev = rows.device
    dyn = dict(d_rows=d_total, rows_mul=rows_mul)
    v, gw = arena.w(pre + "last_layer.weight_v"), arena.w(pre + "last_layer.weight_g")
    K, D = v.shape
    dw = torch.empty((K, D), dtype=F32, device=dev)
    ops.gemm_tn(d_logits, zn, dw, accumulate=False, **dyn)                       # dW_eff = d_logits^T . zn
    ops.weightnorm_bwd(v, gw, winv, dw, arena.g(pre + "last_layer.weight_v"),
                       arena.g(pre + "last_layer.weight_g") if last_layer_trainable_g else None)
    # [rows, D] = d_logits . W: two column tiles x ~26 live row tiles and a 65536-long contraction - cut into slices that add
    # up in fp32 (ccd_gemm_nt, EPI_ATOMIC: 611 -> ~100 us)
    dzn32 = torch.zeros((d_logits.shape[0], D), dtype=F32, device=dev)
    ops.gemm_nt(d_logits, w_t, epilogue=ops.EPI_ATOMIC, out=dzn32, m_fastest=0, **dyn)
    dzn = dzn32.to(BF16)
    dz = torch.zeros_like(z)
    ops.l2norm_bwd(z, inv, dzn, dz, **dyn)
    ops.gemm_tn(dz, a1, arena.g(pre + "mlp.4.weight"), **dyn)
    ops.colsum_bf16(dz, arena.g(pre + "mlp.4.bias"), **dyn)
    du1 = ops.gemm_nt(dz, arena.wbt(pre + "mlp.4.weight"), epilogue=ops.EPI_DGELU, aux=u1, **dyn)
    ops.gemm_tn(du1, a0, arena.g(pre + "mlp.2.weight"), **dyn)
    ops.colsum_bf16(du1, arena.g(pre + "mlp.2.bias"), **dyn)
    du0 = ops.gemm_nt(du1, arena.wbt(pre + "mlp.2.weight"), epilogue=ops.EPI_DGELU, aux=u0, **dyn)
    ops.gemm_tn(du0, rows, arena.g(pre + "mlp.0.weight"), **dyn)
    ops.colsum_bf16(du0, arena.g(pre + "mlp.0.bias"), **dyn)
    d_rows_t = torch.zeros_like(rows)
    ops.gemm_nt(du0, arena.wbt(pre + "mlp.0.weight"), out=d_rows_t, **dyn)
    return d_rows_t


# bf16 logit gradients handed from DinoLossFn.backward to HeadFn.backward, keyed by the logits buffer's address.
# autograd insists that a gradient has its tensor's dtype (fp32 logits), which would cost a bf16 -> fp32 -> bf16 round
# trip over the [rows, 65536] matrix (~1.7 ms per step at B = 256); the loss therefore returns a zero-stride dummy and
# parks the real bf16 gradient here.
_BF16_LOGIT_GRADS = {}


class HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, module, d_total, rows_mul=2, lazy=False):
        save = ctx.needs_input_grad[0]
        logits, saved = head_forward(module.arena, module.arena_prefix, rows, d_total, save, rows_mul, lazy)
        ctx.module, ctx.saved, ctx.d_total, ctx.rows_mul = module, saved, d_total, rows_mul
        if isinstance(logits, LazyLogits):
            # the autograd output is a 1-element stand-in; head_rows() hands the caller the LazyLogits parked under its address
            stub = torch.zeros(1, dtype=F32, device=rows.device)
            _LAZY_LOGITS[stub.data_ptr()] = logits
            logits = stub
        ctx.logits_key = logits.data_ptr()
        ctx.logits_shape = tuple(logits.shape)
        _BF16_LOGIT_GRADS.pop(ctx.logits_key, None)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        m = ctx.module
        parked = _BF16_LOGIT_GRADS.pop(ctx.logits_key, None)
        if parked is not None and tuple(parked.shape) == tuple(d_logits.shape):
            if any(st != 0 for st in d_logits.stride()):          # another consumer contributed a real gradient
                parked = parked + d_logits.to(BF16)
            d_logits = parked
        elif ctx.logits_shape == (1,):                           # lazy logits: the gradient only ever arrives parked
            if parked is None:
                raise RuntimeError("HeadFn.backward: no logit gradient was parked for a head whose logits were left to the fused loss")
            d_logits = parked
        d_rows = head_backward(m.arena, m.arena_prefix, ctx.saved, d_logits.contiguous().to(BF16), ctx.d_total,
                               m.weight_g_trainable, ctx.rows_mul)
        ctx.saved = None
        if m.grad_ready_hook is not None:
            m.grad_ready_hook(m.arena_prefix)
        return d_rows, None, None, None, None


def head_rows(module, rows, d_total, rows_mul=2, lazy=False):
    """HeadFn.apply; with `lazy` the result may be a LazyLogits (its .stub carries the autograd edge)."""
    out = HeadFn.apply(rows, module, d_total, rows_mul, lazy)
    handle = _LAZY_LOGITS.pop(out.data_ptr(), None) if lazy and out.numel() == 1 else None
    if handle is None:
        return out
    handle.stub = out
    return handle


# ---------------------------------------------------------------------------------------------------------- losses
class FusedDinoLossFn(torch.autograd.Function):
    """DinoLossFn on the FACTORS of the two logit matrices (LazyLogits): ccd_head_loss_fwd / _bwd, the logits are never written.
    The student's bf16 logit gradient is parked for HeadFn.backward (see _BF16_LOGIT_GRADS); the stub's own gradient is a zero."""

    @staticmethod
    def forward(ctx, stub, s, t, center, d_total, student_temp, teacher_temp):
        dev = stub.device
        stats = torch.empty((s.zn.shape[0], 4), dtype=F32, device=dev)
        loss = torch.zeros(1, dtype=F32, device=dev)
        center = center.clone()     # DINOLoss.update_center rewrites the buffer in place right after this forward
        ops.head_loss_fwd(s.zn, t.zn, s.w, t.w, center, d_total, student_temp, teacher_temp, stats, loss)
        ctx.saved = (s.zn, t.zn, s.w, t.w, center, stats)
        ctx.d_total, ctx.temps, ctx.key = d_total, (student_temp, teacher_temp), stub.data_ptr()
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        zs, zt, ws, wt, center, stats = ctx.saved
        ctx.saved = None
        d_logits = torch.empty((zs.shape[0], ws.shape[0]), dtype=BF16, device=zs.device)     # rows past 2M are never read
        ops.head_loss_bwd(zs, zt, ws, wt, center, ctx.d_total, ctx.temps[0], ctx.temps[1], stats, 1.0, d_logits,
                          d_grad_scale=d_loss.contiguous().float())
        _BF16_LOGIT_GRADS[ctx.key] = d_logits
        return torch.zeros(1, dtype=F32, device=zs.device), None, None, None, None, None, None


def dino_loss(s_logits, t_logits, center, d_total, student_temp, teacher_temp, park_grad):
    """The distillation loss of Dino_loss.py:81-105 on logits tensors or LazyLogits (both lazy: the fused kernels)."""
    if isinstance(s_logits, LazyLogits) and isinstance(t_logits, LazyLogits) and s_logits.stub is not None and \
            s_logits.shape == t_logits.shape and s_logits.zn.shape == t_logits.zn.shape:
        return FusedDinoLossFn.apply(s_logits.stub, s_logits, t_logits, center, d_total, student_temp, teacher_temp)
    if isinstance(s_logits, LazyLogits):
        raise RuntimeError("a student head's lazy logits cannot be materialised behind autograd's back: run the head with lazy=False")
    t = logits_tensor(t_logits)
    return DinoLossFn.apply(s_logits, t.detach(), center, d_total, student_temp, teacher_temp, park_grad)

class DinoLossFn(torch.autograd.Function):
    """loss (1-element fp32 tensor) of the two cross-view CE terms; d(student logits) in bf16."""

    @staticmethod
    def forward(ctx, s_logits, t_logits, center, d_total, student_temp, teacher_temp, park_grad=False):
        dev = s_logits.device
        ctx.park_grad = park_grad
        stats = torch.empty((s_logits.shape[0], 4), dtype=F32, device=dev)
        loss = torch.zeros(1, dtype=F32, device=dev)
        center = center.clone()     # DINOLoss.update_center rewrites the buffer in place right after this forward
        ops.dino_loss_fwd(s_logits, t_logits, center, d_total, student_temp, teacher_temp, stats, loss)
        ctx.save_for_backward(s_logits, t_logits, center, stats)
        ctx.d_total, ctx.temps = d_total, (student_temp, teacher_temp)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        s_logits, t_logits, center, stats = ctx.saved_tensors
        park = ctx.park_grad and s_logits.dtype == F32
        # parked gradients are only read through the device-side row count: rows past 2M need no zero fill (436 MB)
        d_logits = (torch.empty if park else torch.zeros)(s_logits.shape, dtype=BF16, device=s_logits.device)
        ops.dino_loss_bwd(s_logits, t_logits, center, ctx.d_total, ctx.temps[0], ctx.temps[1], stats, 1.0, d_logits,
                          d_grad_scale=d_loss.contiguous().float())      # upstream gradient stays a device scalar
        if park:                                                         # see _BF16_LOGIT_GRADS
            _BF16_LOGIT_GRADS[s_logits.data_ptr()] = d_logits
            d_logits = torch.zeros((), dtype=F32, device=s_logits.device).expand(s_logits.shape)
        return d_logits, None, None, None, None, None, None


class SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg_logits, mask_a, idmap_b):
        dev = seg_logits.device
        loss = torch.zeros(1, dtype=F32, device=dev)
        d_logits = torch.empty(seg_logits.shape, dtype=F32, device=dev) if ctx.needs_input_grad[0] else None   # NCHW-contiguous
        ops.seg_loss(seg_logits.contiguous(), mask_a, idmap_b, 1.0, loss, d_logits)
        ctx.d_logits = d_logits
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        d = ctx.d_logits
        ctx.d_logits = None
        return d * d_loss, None, None          # 8 MB elementwise at B=256: not worth a kernel argument
