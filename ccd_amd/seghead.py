"""Segmentation head of the student as hand-written HIP (Dino/modules/segmentor.py: MLAHead 38-70, SegHead 73-95).

Activations are channels-last bf16 matrices [pixels, C] - exactly the token-major layout the backbone taps already
have - so every convolution is a GEMM over pixels:

  Conv2d 3x3 / 1x1 (bias=False)      implicit GEMM, A gathered tap by tap (ops.conv_gemm, no im2col buffer)
  ConvTranspose2d(4, 2, 1)           four output-parity classes, each a 2x2-tap implicit GEMM whose epilogue scatters
                                     row (n, y, x) to (n, 2y+py, 2x+px)
  BatchNorm2d (train) + ReLU         batch statistics come out of the GEMM epilogue (column sum / sum of squares),
                                     one elementwise pass normalises; SyncBatchNorm = all-reduce of the [2C] statistics
  Conv2d(128, 2, 3) classifier       N = 2 is no MFMA tile: factored through pixel-wise GEMMs (18 tap/class planes)
                                     plus a 9-point gather-sum; fp32 NCHW logits as the loss expects
  backward                           data gradients: implicit GEMMs with flipped / strided taps; weight gradients:
                                     pixel-contracting TN GEMMs whose B operand is gathered from the image
                                     (ops.conv_wgrad, no patch matrix), re-laid into the parameter layout by
                                     ops.permute4; BN backward = one reduce pass + one apply pass.

Weights are read from the flat parameter arena (fp32 masters) and re-laid into bf16 GEMM operands once per step.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
TAPS3 = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]            # tap = ky*3 + kx, source offset (ky-1, kx-1)
TAPS3_FLIP = [(-dy, -dx) for dy, dx in TAPS3]                            # data gradient reads dY at p - offset
TAPS1 = [(0, 0)]
# ConvTranspose2d(k=4, s=2, p=1): output y = 2*iy - 1 + ky.  Parity class py of the output reads
#   py = 0: ky = 1 (iy = q), ky = 3 (iy = q-1);   py = 1: ky = 0 (iy = q+1), ky = 2 (iy = q)
_KY = {0: ((1, 0), (3, -1)), 1: ((0, 1), (2, 0))}                        # parity -> ((k, source offset), ...)
TAPS_T_GRAD = [(ky - 1, kx - 1) for ky in range(4) for kx in range(4)]   # data gradient: dOut at (2iy - 1 + ky, ...)


def _parity_taps(py, px):
    return [(ky, kx, dy, dx) for ky, dy in _KY[py] for kx, dx in _KY[px]]


def _env_switch(name):
    import os
    v = os.environ.get(name)
    return None if v is None else v not in ("0", "", "false", "off")


# The last level - unpool2's BatchNorm + ReLU and the classifier conv - as the three fused kernels of kernels/cls_tail.h (read ONCE
# at import from CCD_FUSE_CLS_TAIL, a lab switch; tests may assign it).  Default: on where the shape is the reference's.
FUSE_CLS_TAIL = _env_switch("CCD_FUSE_CLS_TAIL")
FORCE_SYNC = False      # 1-GPU smoke of the N > 1 path: SyncBatchNorm layers exchange their statistics on a single rank too


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _BN:
    """State of one BatchNorm2d(+ReLU) application inside a forward pass.  The batch statistics (forward) and the two
    reduction sums of the backward pass live in slices of a `_BNGroup` buffer, so that the SyncBatchNorm exchanges of
    independent layers (the three MLA branches) travel in ONE all-reduce per level and direction: 4 + 4 small collectives
    per step instead of 8 + 8."""

    def __init__(self, mod, rows, device, stats=None):
        self.mod = mod
        self.C = mod.num_features
        self.rows = rows
        self.sync = isinstance(mod, nn.SyncBatchNorm) and (_world() > 1 or (FORCE_SYNC and dist.is_initialized()))
        self.count = float(rows * (_world() if self.sync else 1))
        self.stats = stats if stats is not None else torch.zeros(2 * self.C, dtype=F32, device=device)
        self.mean_rstd = torch.empty(2 * self.C, dtype=F32, device=device)

    def finalize(self):
        """Batch statistics (already summed over the ranks by the group) -> mean / rstd + running statistics."""
        mod = self.mod
        if not mod.training:                          # eval: running statistics (tiny host-side glue, no batch stats)
            self.mean_rstd[: self.C] = mod.running_mean
            self.mean_rstd[self.C:] = torch.rsqrt(mod.running_var + mod.eps)
            return
        momentum = mod.momentum if mod.momentum is not None else 1.0 / float(mod.num_batches_tracked.item() + 1)
        ops.bn_finalize(self.stats, self.count, mod.eps, momentum, self.mean_rstd, mod.running_mean, mod.running_var)
        mod.num_batches_tracked += 1

    def forward(self, x, out):
        return ops.bn_relu_fwd(x, self.mean_rstd, self.mod.weight, self.mod.bias, out)

    def backward_reduce(self, dy, x, red):
        """red [2C] (a group slice, zeroed) += the two sums BatchNorm's input gradient needs."""
        ops.bn_relu_bwd_reduce(dy, x, self.mean_rstd, self.mod.weight, self.mod.bias, red)

    def backward_apply(self, dy, x, dx, red, red_local):
        """dx (may alias dy) = gradient w.r.t. the BN input; parameter gradients (from the LOCAL sums) accumulate into .grad."""
        mod = self.mod
        if not mod.training:                          # eval-mode BN is an affine map: no batch terms
            red = torch.zeros_like(red)
        return ops.bn_relu_bwd_apply(dy, x, self.mean_rstd, mod.weight, mod.bias, red, self.count, red_local,
                                     mod.weight.grad, mod.bias.grad, dx)


class _BNGroup:
    """BatchNorm layers of one level whose statistics are exchanged together."""

    def __init__(self, mods, rows, device, buf=None):
        sizes = [2 * m.num_features for m in mods]
        self.buf = buf if buf is not None else torch.zeros(sum(sizes), dtype=F32, device=device)      # (buf: a zeroed slice of the caller's arena)
        assert self.buf.numel() == sum(sizes)
        offs = [sum(sizes[:i]) for i in range(len(sizes))]
        self.bns = [_BN(m, rows, device, self.buf[o:o + n]) for m, o, n in zip(mods, offs, sizes)]
        self.sync = any(b.sync for b in self.bns)
        self._offs, self._sizes = offs, sizes

    def finalize(self):
        if self.sync and self.bns[0].mod.training:
            dist.all_reduce(self.buf)                 # one collective for the whole level
        if all(b.mod.training and b.mod.momentum is not None for b in self.bns) and len(self.bns) <= 4:
            # one launch for the level: mean / rstd, running statistics and the batch counters of all its layers
            ops.bn_finalize_multi([(b.stats, b.count, b.mod.eps, b.mod.momentum, b.mean_rstd, b.mod.running_mean, b.mod.running_var,
                                    b.mod.num_batches_tracked) for b in self.bns])
            return
        for b in self.bns:
            b.finalize()

    def backward(self, dys, xs, dxs, red=None):
        """dys / xs / dxs: per layer.  Reduce all layers, ONE all-reduce, apply all layers.  red: a zeroed slice of the caller's arena."""
        red = red if red is not None else torch.zeros_like(self.buf)
        slices = [red[o:o + n] for o, n in zip(self._offs, self._sizes)]
        for b, dy, x, r in zip(self.bns, dys, xs, slices):
            b.backward_reduce(dy, x, r)
        red_local = red
        if self.sync:
            red_local = red.clone()
            dist.all_reduce(red)
        loc = [red_local[o:o + n] for o, n in zip(self._offs, self._sizes)]
        return [b.backward_apply(dy, x, dx, r, rl) for b, dy, x, dx, r, rl in zip(self.bns, dys, xs, dxs, slices, loc)]

    def backward_cls_tail(self, d_logits, y, cls, dbias_t, images, H, W, red=None):
        """The group of the LAST level (one BatchNorm) under the classifier `cls`: kernels/cls_tail.h.  -> d(y) bf16; cls.weight.grad,
        cls.bias.grad, the BatchNorm's parameter gradients and dbias_t (the transposed conv's bias gradient) accumulate."""
        bn = self.bns[0]
        mod = bn.mod
        w = cls.weight.detach()
        red = red if red is not None else torch.zeros_like(self.buf)
        ops.cls_tail_bwd_reduce(d_logits, y, bn.mean_rstd, mod.weight.detach(), mod.bias.detach(), w, red, cls.bias.grad, images, H, W)
        red_local = red
        if self.sync:
            red_local = red.clone()
            dist.all_reduce(red)
        if not mod.training:                              # eval-mode BN is an affine map: no batch terms
            red = torch.zeros_like(red)
        return ops.cls_tail_bwd_apply(d_logits, y, bn.mean_rstd, mod.weight.detach(), mod.bias.detach(), w, red, bn.count, red_local,
                                      mod.weight.grad, mod.bias.grad, cls.weight.grad, dbias_t, torch.empty_like(y), images, H, W)


class _Zeros:
    """One zero-filled fp32 allocation handed out in slices: the head's statistics / reduction / staging buffers were a dozen
    `torch.zeros` (a 5-us fill launch each) per pass."""

    def __init__(self, sizes, device):
        self.buf = torch.zeros(sum(-(-n // 64) * 64 for n in sizes), dtype=F32, device=device)       # (256-byte aligned slices)
        self.at = 0

    def take(self, n):
        out = self.buf[self.at:self.at + n]
        self.at += -(-n // 64) * 64
        assert self.at <= self.buf.numel()
        return out


def _ensure_grads(module):
    for p in module.parameters():
        if p.requires_grad and p.grad is None:
            p.grad = torch.zeros_like(p)


def cls_forward(x, w, b, images, H, W):
    """Conv2d(C, 2, 3, padding=1) on channels-last bf16 x [pixels, C] -> fp32 logits [images, 2, H, W]."""
    C = x.shape[1]
    wz = torch.zeros((32, C), dtype=BF16, device=x.device)                       # row co*9+tap = w[co, :, tap]
    ops.permute4(w, (C * 9, 1, 9), (2, 9, C), wz[:18])
    zT = ops.gemm_nt(wz, x, epilogue=ops.EPI_F32)                                 # [32, pixels] fp32
    return ops.cls_gather_fwd(zT, b, images, H, W)


def cls_backward(d_logits, x, w, dw, db, images, H, W):
    """-> dx bf16 [pixels, C]; dw [2, C, 3, 3] / db [2] accumulate (fp32)."""
    C, dev = x.shape[1], x.device
    g = ops.cls_grad_cols(d_logits, images, H, W)                                 # [pixels, 64], col co*9+tap
    wd = torch.zeros((C, 64), dtype=BF16, device=dev)                             # wd[c, co*9+tap] = w[co, c, tap]
    ops.permute4(w, (9, C * 9, 1), (C, 2, 9), wd, dst_strides=(64, 9, 1))
    dx = ops.gemm_nt(g, wd)
    stage = torch.zeros((64, C), dtype=F32, device=dev)
    ops.gemm_tn(g, x, stage)                                                      # [co*9+tap, c]
    ops.permute4(stage, (9 * C, 1, C), (2, C, 9), dw, accumulate=True)
    cs = torch.zeros(64, dtype=F32, device=dev)
    ops.colsum_bf16(g, cs)                                                        # centre tap (always in bounds)
    ops.permute4(cs[4:], (9,), (2,), db, accumulate=True)
    return dx


def _relaid_weights(head, E, mid, out_c, dev, backward=True):
    """Every bf16 GEMM operand the head's forward AND (backward=True) backward take from its fp32 master weights, re-laid in ONE
    launch (ops.permute4_multi; they were 22 launches of ~5 us per step).  -> dict of tensors.  The backward's operands (w2t, w1d,
    wt) are a snapshot taken at forward time - the weights the forward pass multiplied by, which is what autograd's saved tensors
    would hold too; a no-grad / eval pass does not build them."""
    heads = [head.mlahead.head2, head.mlahead.head3, head.mlahead.head4]
    w, jobs = {"w1": [], "w2": [], "w2t": [], "w1d": [], "wp": [], "wt": []}, []

    def job(key, src, strides, dims, shape):
        dst = torch.empty(shape, dtype=BF16, device=dev)
        jobs.append((src, strides, dims, dst))
        w[key].append(dst)

    for seq in heads:
        w3, w1x1 = seq[0].weight.detach(), seq[3].weight.detach()
        job("w1", w3, (E * 9, 1, 9), (mid, 9, E), (mid, 9 * E))                    # forward 3x3: [co][tap][ci]
        job("w2", w1x1, (mid, 1), (out_c, mid), (out_c, mid))                      # forward 1x1
        if backward:
            job("w2t", w1x1, (1, mid), (mid, out_c), (mid, out_c))                 # its data gradient
            job("w1d", w3, (9, 1, E * 9), (E, 9, mid), (E, 9 * mid))               # 3x3 data gradient: [ci][tap][co] <- W[co][ci][tap]
    for seq in (head.unpool1, head.unpool2):
        convt = seq[0]
        cin, cout = convt.in_channels, convt.out_channels
        wsrc = convt.weight.detach()                                               # [cin, cout, 4, 4]
        per = []
        for py in (0, 1):
            for px in (0, 1):
                pt = _parity_taps(py, px)
                ky0, kx0 = pt[0][0], pt[0][1]                                      # taps advance by +2 in ky (outer) / kx (inner)
                dst = torch.empty((cout, 4 * cin), dtype=BF16, device=dev)
                jobs.append((wsrc.reshape(-1)[ky0 * 4 + kx0:], (16, 8, 2, cout * 16), (cout, 2, 2, cin), dst))
                per.append(dst)
        w["wp"].append(per)
        if backward:
            job("wt", wsrc, (cout * 16, 1, 16), (cin, 16, cout), (cin, 16 * cout)) # data gradient: [ci][tap][co] <- W[ci][co][tap]
    ops.permute4_multi(jobs)
    return w


class SegHeadFn(torch.autograd.Function):
    """(tap2, tap3, tap4: bf16 [N*256, E]) -> fp32 logits [N, 2, 32, 128]; parameter gradients go straight to .grad."""

    @staticmethod
    def forward(ctx, head, images, t2, t3, t4):
        dev = t2.device
        taps = [t.contiguous() for t in (t2, t3, t4)]
        E = taps[0].shape[1]
        gh, gw = 8, 32
        M = images * gh * gw
        assert taps[0].shape[0] == M
        mla = head.mlahead
        heads = [mla.head2, mla.head3, mla.head4]
        mid, out_c = heads[0][0].out_channels, heads[0][3].out_channels
        ups_mods = (head.unpool1, head.unpool2)
        d3 = ops.conv_desc((gh, gw), (gh, gw), E, TAPS3)
        d1 = ops.conv_desc((gh, gw), (gh, gw), mid, TAPS1)
        cat = torch.empty((M, 3 * out_c), dtype=BF16, device=dev)
        wts = _relaid_weights(head, E, mid, out_c, dev, backward=any(ctx.needs_input_grad))
        zeros = _Zeros([6 * mid, 6 * out_c] + [2 * seq[0].out_channels for seq in ups_mods], dev)     # the four levels' statistics
        saved = {"taps": taps, "y1": [], "a1": [], "y2": [], "w": wts}
        # level 1 of the three independent branches (3x3 conv), ONE statistics exchange, then level 2 (1x1 conv), ONE more
        g1 = _BNGroup([seq[1] for seq in heads], M, dev, zeros.take(6 * mid))
        for i, seq in enumerate(heads):
            y1 = torch.empty((M, mid), dtype=BF16, device=dev)
            ops.conv_gemm(taps[i], d3, wts["w1"][i], M, y1, colsum=g1.bns[i].stats[:mid], colsumsq=g1.bns[i].stats[mid:])
            saved["y1"].append(y1)
        g1.finalize()
        g2 = _BNGroup([seq[4] for seq in heads], M, dev, zeros.take(6 * out_c))
        for i, seq in enumerate(heads):
            a1 = g1.bns[i].forward(saved["y1"][i], torch.empty_like(saved["y1"][i]))
            y2 = torch.empty((M, out_c), dtype=BF16, device=dev)
            ops.conv_gemm(a1, d1, wts["w2"][i], M, y2, colsum=g2.bns[i].stats[:out_c], colsumsq=g2.bns[i].stats[out_c:])
            for k, v in (("a1", a1), ("y2", y2)):
                saved[k].append(v)
        g2.finalize()
        for i in range(3):
            g2.bns[i].forward(saved["y2"][i], cat[:, i * out_c:(i + 1) * out_c])
        saved["g1"], saved["g2"] = g1, g2
        x, grid = cat, (gh, gw)
        ups = []
        for lvl, seq in enumerate(ups_mods):
            convt, bnm = seq[0], seq[1]
            cin, cout = convt.in_channels, convt.out_channels
            rows = images * grid[0] * grid[1]
            y = torch.empty((4 * rows, cout), dtype=BF16, device=dev)
            grp = _BNGroup([bnm], 4 * rows, dev, zeros.take(2 * cout))
            bn = grp.bns[0]
            for cls_i, (py, px) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                pt = _parity_taps(py, px)
                desc = ops.conv_desc(grid, grid, cin, [(dy, dx) for _, _, dy, dx in pt], parity=(py, px))
                ops.conv_gemm(x, desc, wts["wp"][lvl][cls_i], rows, y, bias=convt.bias, colsum=bn.stats[:cout],
                              colsumsq=bn.stats[cout:])
            grp.finalize()
            ups.append((x, y, grp, grid))
            grid = (2 * grid[0], 2 * grid[1])
            last = seq is head.unpool2
            fused_tail = (last and FUSE_CLS_TAIL is not False and head.cls.out_channels == 2
                          and ops.cls_tail_supported(y, grid[0], grid[1]))
            # (the last level's relu(bn(y)) is consumed by the classifier only: formed in its kernel's registers, never written)
            x = None if fused_tail else bn.forward(y, torch.empty_like(y))
        if fused_tail:
            logits = ops.cls_tail_fwd(y, bn.mean_rstd, bn.mod.weight.detach(), bn.mod.bias.detach(), head.cls.weight.detach(),
                                      head.cls.bias.detach(), images, grid[0], grid[1])
        else:
            logits = cls_forward(x, head.cls.weight.detach(), head.cls.bias.detach(), images, grid[0], grid[1])
        ctx.head, ctx.images, ctx.saved, ctx.ups, ctx.a_last, ctx.grid = head, images, saved, ups, x, grid
        ctx.fused_tail = fused_tail
        ctx.dims = (E, mid, out_c, M)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        head, images, saved = ctx.head, ctx.images, ctx.saved
        E, mid, out_c, M = ctx.dims
        wts = saved["w"]
        dev = d_logits.device
        _ensure_grads(head.mlahead)
        for m in (head.unpool1, head.unpool2, head.cls):
            _ensure_grads(m)
        H, W = ctx.grid
        ups_mods = (head.unpool2, head.unpool1)
        heads = [head.mlahead.head2, head.mlahead.head3, head.mlahead.head4]
        # every zero-initialised buffer of the pass from one fill: the levels' reduction sums, then the staged weight gradients
        stage_sizes = [seq[0].in_channels * 16 * seq[0].out_channels for seq in ups_mods] + [mid * 9 * E] * 3
        zeros = _Zeros([2 * seq[0].out_channels for seq in ups_mods] + [6 * out_c, 6 * mid] + stage_sizes, dev)
        reds = [zeros.take(2 * seq[0].out_channels) for seq in ups_mods] + [zeros.take(6 * out_c), zeros.take(6 * mid)]
        folds = []                                      # (stage, strides, dims, grad): folded back into the parameter layout in ONE launch
        d_logits = d_logits.contiguous().float()
        d = None
        if not ctx.fused_tail:
            d = cls_backward(d_logits, ctx.a_last, head.cls.weight.detach(), head.cls.weight.grad, head.cls.bias.grad, images, H, W)
        ctx.a_last = None
        # ---- transposed convs, last first
        for lvl, (seq, (x_in, y, grp, grid)) in enumerate(zip(ups_mods, reversed(ctx.ups))):
            convt = seq[0]
            cin, cout = convt.in_channels, convt.out_channels
            rows = images * grid[0] * grid[1]
            if d is None:                                                           # classifier + BatchNorm backward, fused
                dyc = grp.backward_cls_tail(d_logits, y, head.cls, convt.bias.grad, images, H, W, red=reds[lvl])
            else:
                dyc = grp.backward([d], [y], [d], red=reds[lvl])[0]                 # in place: d(convT output)
                ops.colsum_bf16(dyc, convt.bias.grad)
            desc = ops.conv_desc(grid, (2 * grid[0], 2 * grid[1]), cout, TAPS_T_GRAD, s_mul=2)
            stage = zeros.take(cin * 16 * cout).view(cin, 16 * cout)
            ops.conv_wgrad(x_in, dyc, desc, stage)                                  # [ci][tap][co]
            folds.append((stage, (16 * cout, 1, cout), (cin, cout, 16), convt.weight.grad))
            d = ops.conv_gemm(dyc, desc, wts["wt"][1 - lvl], rows, torch.empty((rows, cin), dtype=BF16, device=dev))
        # ---- the three 3x3 -> 1x1 branches; d = gradient of the concatenated [M, 3*out_c] map
        gh, gw = 8, 32
        d3 = ops.conv_desc((gh, gw), (gh, gw), E, TAPS3)
        d3f = ops.conv_desc((gh, gw), (gh, gw), mid, TAPS3_FLIP)
        # level 2 of all three branches (one exchange), their 1x1 products, level 1 (one exchange), their 3x3 products
        dy2s = saved["g2"].backward([d[:, i * out_c:(i + 1) * out_c] for i in range(3)], saved["y2"],
                                    [torch.empty_like(y) for y in saved["y2"]], red=reds[2])
        da1s = []
        for i, seq in enumerate(heads):
            ops.gemm_tn(dy2s[i], saved["a1"][i], seq[3].weight.grad.view(out_c, mid))  # dW2[co][ci]
            da1s.append(ops.gemm_nt(dy2s[i], wts["w2t"][i]))                        # [M, mid]
        dy1s = saved["g1"].backward(da1s, saved["y1"], da1s, red=reds[3])
        d_taps = []
        for i, seq in enumerate(heads):
            stage = zeros.take(mid * 9 * E).view(mid, 9 * E)
            ops.conv_wgrad(dy1s[i], saved["taps"][i], d3, stage)                    # [co][tap][ci]
            folds.append((stage, (9 * E, 1, E), (mid, E, 9), seq[0].weight.grad))
            d_taps.append(ops.conv_gemm(dy1s[i], d3f, wts["w1d"][i], M, torch.empty((M, E), dtype=BF16, device=dev)))
        ops.permute4_multi(folds, accumulate=True)
        ctx.saved = ctx.ups = None
        return (None, None, *d_taps)


def seg_head_forward(head, taps, images):
    """taps: three bf16 [images*256, E] token-major feature maps -> fp32 logits [images, 2, 32, 128]."""
    return SegHeadFn.apply(head, images, *taps)
