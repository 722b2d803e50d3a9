"""Thin tensor-level wrappers over the C ABI (include/ccd_hip.h).  Every function enqueues HIP kernels on the
current stream; nothing here computes on the host."""
from __future__ import annotations

import ctypes

import torch

from . import _lib

EPI_BF16, EPI_GELU, EPI_RESID, EPI_F32, EPI_ATOMIC, EPI_DGELU = range(6)
BF16, F32 = torch.bfloat16, torch.float32
_EPI_NAMES = ("bf16", "gelu", "resid", "f32", "atomic", "dgelu")


class KernelTimer:
    """Optional per-launch timing with HIP events on the launching stream (bench.py's roofline leg).  Events are
    only read after the timed region has been synchronised, so recording them never stalls the stream."""

    def __init__(self, only=None):
        self.records = []          # (key, flops, algorithmic bytes, start_event, end_event)
        self.only = only           # optional set of keys: every other launch runs without events

    def span(self, key, flops, nbytes=0.0):
        if self.only is not None and key not in self.only:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((key, flops, nbytes, e0, e1))
        return e0, e1

    def summary(self):
        out = {}
        for key, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


TIMER = None     # set to a KernelTimer to time every GEMM / attention / LayerNorm / loss launch


class _Span:
    """with _Span(key, flops, algorithmic bytes): the launches inside are bracketed by HIP events when a KernelTimer is attached."""

    def __init__(self, key, flops, nbytes):
        self.ev = TIMER.span(key, flops, nbytes) if TIMER is not None else None

    def __enter__(self):
        if self.ev:
            self.ev[0].record()

    def __exit__(self, *a):
        if self.ev:
            self.ev[1].record()


def policy_set(key, value):
    """Kernel-selection policy of the library (include/ccd_hip.h: ccd_policy_set), e.g. policy_set('gemm_256_min_m', 1)."""
    _lib.check(_lib.get().ccd_policy_set(key.encode(), int(value)), f"policy_set({key})")


def policy_get(key):
    v = ctypes.c_int(0)
    _lib.check(_lib.get().ccd_policy_get(key.encode(), ctypes.byref(v)), f"policy_get({key})")
    return v.value


class policy:
    """with ops.policy(gemm_256_min_m=1, gemm_row384=2): ...   (restores the previous values on exit)"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.saved = {k: policy_get(k) for k in self.kw}
        for k, v in self.kw.items():
            policy_set(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.saved.items():
            policy_set(k, v)


def _chk(t, dtype, name):
    if t is None:
        return
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def gemm_nt(a, b, *, epilogue=EPI_BF16, out=None, out2=None, bias=None, resid=None, rowscale=None,
            rows_per_sample=1, aux=None, alpha=1.0, m_fastest=None, d_rows=None, rows_mul=1, colsum=None,
            store_u=True):
    """out[M,N] = a[M,K] @ b[N,K]^T with a fused epilogue (see include/ccd_hip.h)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(bias, F32, "bias"); _chk(resid, F32, "resid")
    _chk(rowscale, F32, "rowscale"); _chk(aux, BF16, "aux")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    odt = F32 if epilogue in (EPI_RESID, EPI_F32, EPI_ATOMIC) else BF16
    if out is None and (store_u or epilogue != EPI_GELU):
        out = torch.empty((M, N), dtype=odt, device=a.device)
    _chk(out, odt, "out")
    if epilogue == EPI_GELU and out2 is None:
        out2 = torch.empty((M, N), dtype=BF16, device=a.device)
    if m_fastest is None:
        m_fastest = 1 if N > M else 0
    lib = _lib.get()
    span = None
    if TIMER is not None and d_rows is None:
        # algorithmic bytes: each operand read once, each output written once (bf16 = 2 B, fp32 = 4 B)
        nbytes = 2.0 * (M * K + N * K) + M * N * {EPI_BF16: 2, EPI_GELU: 2 + (2 if store_u else 0), EPI_RESID: 8, EPI_F32: 4,
                                                   EPI_ATOMIC: 8, EPI_DGELU: 4}[epilogue]
        span = TIMER.span("gemm_nt_" + _EPI_NAMES[epilogue], 2.0 * M * N * K, nbytes)
    if span:
        span[0].record()
    _lib.check(lib.ccd_gemm_nt(_lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), M, N, K, epilogue, _lib.ptr(out),
                               N if out is None else out.stride(0), _lib.ptr(out2), 0 if out2 is None else out2.stride(0), _lib.ptr(bias),
                               _lib.ptr(resid), 0 if resid is None else resid.stride(0), _lib.ptr(rowscale),
                               rows_per_sample, _lib.ptr(aux), 0 if aux is None else aux.stride(0), float(alpha),
                               int(m_fastest), _lib.ptr(d_rows), int(rows_mul), _lib.ptr(colsum), _lib.stream()),
               "gemm_nt")
    if span:
        span[1].record()
    return (out, out2) if epilogue == EPI_GELU else out      # (EPI_DGELU: out2, when given, receives gelu(aux))


def gemm_nt_resid_ln(a, b, *, bias, resid, rowscale, rows_per_sample, gamma, beta, eps, out=None):
    """out (fp32) = resid + (a @ b^T + bias) * rowscale[row // rows_per_sample];  y = LayerNorm(out) * gamma + beta.
    -> (out, y bf16, mean, rstd): the residual product with the following LayerNorm folded into its epilogue."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(bias, F32, "bias"); _chk(resid, F32, "resid"); _chk(rowscale, F32, "rowscale")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and N <= 512
    if out is None:
        out = torch.empty((M, N), dtype=F32, device=a.device)
    y = torch.empty((M, N), dtype=BF16, device=a.device)
    mean = torch.empty(M, dtype=F32, device=a.device)
    rstd = torch.empty(M, dtype=F32, device=a.device)
    span = TIMER.span("gemm_nt_resid", 2.0 * M * N * K, 2.0 * (M * K + N * K) + 10.0 * M * N) if TIMER is not None else None
    if span:
        span[0].record()
    _call("ccd_gemm_nt_resid_ln", _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), M, N, K, _lib.ptr(out), out.stride(0),
          _lib.ptr(bias), _lib.ptr(resid), resid.stride(0), _lib.ptr(rowscale), int(rows_per_sample), _lib.ptr(gamma),
          _lib.ptr(beta), float(eps), _lib.ptr(y), y.stride(0), _lib.ptr(mean), _lib.ptr(rstd))
    if span:
        span[1].record()
    return out, y, mean, rstd


def lnbwd_tap_supported(g, N):
    """ccd_gemm_nt_lnbwd_tap_g16 takes the bf16 gradient stream at N = 384 (the ViT-Small path)."""
    return g.dtype == BF16 and N == 384 and policy_get("rowgemm") != 0


def gemm_nt_lnbwd(a, b, x, mean, rstd, gamma, g, dgamma, dbeta, accumulate=True, gb=None, rowscale=None, rows_per_sample=1,
                  dbias=None, tap=None):
    """dy = a @ b^T is the gradient of a LayerNorm output: g (+)= LN'(dy) with dgamma / dbeta (+ the bf16 tail of ln_bwd) in
    the epilogue of the product - dy never reaches HBM (ccd_gemm_nt + ccd_ln_bwd in one launch; N <= 384 or N = 512).
    tap = (d_tap bf16 [M, N], tap_gamma, tap_dgamma, tap_dbeta): a second LayerNorm of the same rows (a segmentation tap: same
    statistics, other gamma) whose backward pass joins the epilogue (lnbwd_tap_supported)."""
    g16 = g.dtype == BF16           # (round 6) the residual-gradient stream in bf16: ccd_gemm_nt_lnbwd_g16
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(x, F32, "x"); _chk(g, BF16 if g16 else F32, "g"); _chk(gb, BF16, "gb")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and tuple(x.shape) == (M, N) and tuple(g.shape) == (M, N) and N <= 512
    gbytes = 2.0 if g16 else 4.0
    span = TIMER.span("gemm_nt_lnbwd", 2.0 * M * N * K, 2.0 * (M * K + N * K) + M * N * (4.0 + (2 * gbytes if accumulate else gbytes))
                      + (2.0 * M * N if gb is not None else 0.0)) if TIMER is not None else None
    if span:
        span[0].record()
    common = (_lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), M, N, K, _lib.ptr(x), x.stride(0),
              _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(g), g.stride(0), 1 if accumulate else 0, _lib.ptr(dgamma),
              _lib.ptr(dbeta), _lib.ptr(gb), 0 if gb is None else gb.stride(0), _lib.ptr(rowscale), int(rows_per_sample),
              _lib.ptr(dbias))
    if tap is not None:
        d_tap, tap_gamma, tap_dgamma, tap_dbeta = tap
        _chk(d_tap, BF16, "d_tap")
        assert g16 and tuple(d_tap.shape) == (M, N)
        _call("ccd_gemm_nt_lnbwd_tap_g16", *common, _lib.ptr(d_tap), d_tap.stride(0), _lib.ptr(tap_gamma), _lib.ptr(tap_dgamma),
              _lib.ptr(tap_dbeta))
    else:
        _call("ccd_gemm_nt_lnbwd_g16" if g16 else "ccd_gemm_nt_lnbwd", *common)
    if span:
        span[1].record()
    return g


def mlp_bwd_fused_supported(g, E, H):
    """ccd_mlp_bwd_fused takes the bf16 gradient stream at E in {256, 384} with H a multiple of 128."""
    return g.dtype == BF16 and E in (256, 384) and H % 128 == 0


def mlp_bwd_fused(gb, w2t, w1t, u, *, db1, x, mean, rstd, gamma, g, dgamma, dbeta, gb_out, rowscale=None, rows_per_sample=1,
                  dbias=None, accumulate=True):
    """The data-gradient chain of the MLP branch in one launch (include/ccd_hip.h: ccd_mlp_bwd_fused):
    du = (gb @ w2t^T) * gelu'(u), dy2 = du @ w1t^T, then LayerNorm-2's backward of dy2 as in gemm_nt_lnbwd (bf16 stream g).
    w2t = fc2.weight^T [H, E], w1t = fc1.weight^T [E, H].  -> du bf16 [M, H]; db1 += colsum(du)."""
    _chk(gb, BF16, "gb"); _chk(w2t, BF16, "w2t"); _chk(w1t, BF16, "w1t"); _chk(u, BF16, "u"); _chk(x, F32, "x"); _chk(g, BF16, "g")
    _chk(gb_out, BF16, "gb_out"); _chk(db1, F32, "db1"); _chk(rowscale, F32, "rowscale")
    M, E = gb.shape
    H = w2t.shape[0]
    assert tuple(w2t.shape) == (H, E) and tuple(w1t.shape) == (E, H) and tuple(u.shape) == (M, H) and tuple(x.shape) == (M, E)
    assert gb_out is None or gb_out.data_ptr() != gb.data_ptr()
    du = torch.empty((M, H), dtype=BF16, device=gb.device)
    # algorithmic bytes: gb, u in; du out; x in; g in + out; gb_out out; the weights once
    nbytes = M * (2.0 * E + 4.0 * H + 4.0 * E + (4.0 if accumulate else 2.0) * E + (2.0 * E if gb_out is not None else 0.0)) + 4.0 * E * H
    span = TIMER.span("mlp_bwd_fused", 4.0 * M * E * H, nbytes) if TIMER is not None else None
    if span:
        span[0].record()
    _call("ccd_mlp_bwd_fused", _lib.ptr(gb), gb.stride(0), _lib.ptr(w2t), w2t.stride(0), _lib.ptr(w1t), w1t.stride(0),
          _lib.ptr(u), u.stride(0), _lib.ptr(du), du.stride(0), _lib.ptr(db1),
          _lib.ptr(x), x.stride(0), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(g), g.stride(0),
          1 if accumulate else 0, _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(gb_out), 0 if gb_out is None else gb_out.stride(0),
          _lib.ptr(rowscale), int(rows_per_sample), _lib.ptr(dbias), M, E, H)
    if span:
        span[1].record()
    return du


def mlp_fused(y, w1, b1, w2, b2, *, resid, rowscale, rows_per_sample, gamma, beta, eps, store_u=False, store_gact=False, out=None):
    """out (fp32) = resid + (gelu(y @ w1^T + b1) @ w2^T + b2) * rowscale[row // rows_per_sample];
    y_next = LayerNorm(out) * gamma + beta  ->  (out, y_next bf16, mean, rstd, u bf16 | None[, gelu(u) bf16 with store_gact]).
    The hidden activation never reaches HBM; `u` (the bf16 pre-activation) is written only when store_u, gelu(u) only when
    store_gact (the weight-gradient product of the backward pass reads it)."""
    _chk(y, BF16, "y"); _chk(w1, BF16, "w1"); _chk(w2, BF16, "w2"); _chk(b1, F32, "b1"); _chk(b2, F32, "b2")
    _chk(resid, F32, "resid"); _chk(rowscale, F32, "rowscale")
    M, E = y.shape
    H = w1.shape[0]
    assert tuple(w1.shape) == (H, E) and tuple(w2.shape) == (E, H) and tuple(resid.shape) == (M, E)
    dev = y.device
    if out is None:
        out = torch.empty((M, E), dtype=F32, device=dev)
    yn = torch.empty((M, E), dtype=BF16, device=dev)
    mean = torch.empty(M, dtype=F32, device=dev)
    rstd = torch.empty(M, dtype=F32, device=dev)
    u = torch.empty((M, H), dtype=BF16, device=dev) if store_u else None
    gact = torch.empty((M, H), dtype=BF16, device=dev) if (store_u and store_gact) else None
    # algorithmic bytes: y read, resid read, out + y_next written (+ u, + gelu(u)), weights once
    nbytes = M * E * (2.0 + 4.0 + 4.0 + 2.0) + (2.0 * M * H if store_u else 0.0) + (2.0 * M * H if gact is not None else 0.0) + 4.0 * E * H
    span = TIMER.span("mlp_fused", 4.0 * M * E * H, nbytes) if TIMER is not None else None
    if span:
        span[0].record()
    _call("ccd_mlp_fused", _lib.ptr(y), y.stride(0), _lib.ptr(w1), w1.stride(0), _lib.ptr(b1), _lib.ptr(w2), w2.stride(0),
          _lib.ptr(b2), _lib.ptr(resid), resid.stride(0), _lib.ptr(rowscale), int(rows_per_sample), _lib.ptr(out),
          out.stride(0), _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(yn), yn.stride(0), _lib.ptr(mean),
          _lib.ptr(rstd), _lib.ptr(u), 0 if u is None else u.stride(0), _lib.ptr(gact), 0 if gact is None else gact.stride(0),
          M, E, H)
    if span:
        span[1].record()
    return (out, yn, mean, rstd, u, gact) if store_gact else (out, yn, mean, rstd, u)


def proj_mlp_fused(a, wp, bp, *, resid, rowscale1, gamma2, beta2, w1, b1, w2, b2, rowscale2, rows_per_sample, gamma, beta, eps,
                   save=False, out=None, tap_gamma=None, tap_beta=None, store_gact=False):
    """The second half of a transformer block in one launch (include/ccd_hip.h: ccd_proj_mlp_fused):
        x_mid = resid + (a @ wp^T + bp) * rowscale1;   y2 = LayerNorm(x_mid) * gamma2 + beta2
        out   = x_mid + (gelu(y2 @ w1^T + b1) @ w2^T + b2) * rowscale2;   y_next = LayerNorm(out) * gamma + beta
    -> (out, y_next, mean, rstd, saved) with saved = (x_mid, y2, mean2, rstd2, u) when `save` (what the backward pass reads), else None:
    then x_mid and y2 never reach HBM.  With tap_gamma / tap_beta a sixth result: LayerNorm(out) * tap_gamma + tap_beta (bf16).
    store_gact (with save): saved gains a sixth member, gelu(u) bf16 [M, H] (ccd_proj_mlp_fused_gact: what ccd_mlp_bwd_fused's caller
    hands to the weight-gradient pair).  Raises RuntimeError('unsupported shape') where the kernel does not apply."""
    _chk(a, BF16, "a"); _chk(wp, BF16, "wp"); _chk(w1, BF16, "w1"); _chk(w2, BF16, "w2"); _chk(resid, F32, "resid")
    _chk(rowscale1, F32, "rowscale1"); _chk(rowscale2, F32, "rowscale2")
    M, E = a.shape
    H = w1.shape[0]
    assert tuple(wp.shape) == (E, E) and tuple(w1.shape) == (H, E) and tuple(w2.shape) == (E, H) and tuple(resid.shape) == (M, E)
    dev = a.device
    if out is None:
        out = torch.empty((M, E), dtype=F32, device=dev)
    yn = torch.empty((M, E), dtype=BF16, device=dev)
    mean = torch.empty(M, dtype=F32, device=dev)
    rstd = torch.empty(M, dtype=F32, device=dev)
    xmid = y2 = mean2 = rstd2 = u = None
    if save:
        xmid = torch.empty((M, E), dtype=F32, device=dev)
        y2 = torch.empty((M, E), dtype=BF16, device=dev)
        mean2 = torch.empty(M, dtype=F32, device=dev)
        rstd2 = torch.empty(M, dtype=F32, device=dev)
        u = torch.empty((M, H), dtype=BF16, device=dev)
    gact = torch.empty((M, H), dtype=BF16, device=dev) if (save and store_gact) else None
    tap = torch.empty((M, E), dtype=BF16, device=dev) if tap_gamma is not None else None
    # algorithmic bytes: a and resid read, out + y_next written (+ x_mid, y2, u [, gelu(u)] when saved), the three weight matrices once
    nbytes = M * E * (2.0 + 4.0 + 4.0 + 2.0) + (M * E * 6.0 + 2.0 * M * H if save else 0.0) + 4.0 * E * H + 2.0 * E * E + \
        (2.0 * M * E if tap is not None else 0.0) + (2.0 * M * H if gact is not None else 0.0)
    span = TIMER.span("proj_mlp_fused", 4.0 * M * E * H + 2.0 * M * E * E, nbytes) if TIMER is not None else None
    if span:
        span[0].record()
    head = (_lib.ptr(a), a.stride(0), _lib.ptr(wp), wp.stride(0), _lib.ptr(bp), _lib.ptr(resid), resid.stride(0),
            _lib.ptr(rowscale1), _lib.ptr(gamma2), _lib.ptr(beta2), _lib.ptr(xmid), 0 if xmid is None else xmid.stride(0), _lib.ptr(y2),
            0 if y2 is None else y2.stride(0), _lib.ptr(mean2), _lib.ptr(rstd2), _lib.ptr(w1), w1.stride(0), _lib.ptr(b1), _lib.ptr(w2),
            w2.stride(0), _lib.ptr(b2), _lib.ptr(rowscale2), int(rows_per_sample), _lib.ptr(out), out.stride(0), _lib.ptr(gamma),
            _lib.ptr(beta), float(eps), _lib.ptr(yn), yn.stride(0), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(u),
            0 if u is None else u.stride(0))
    tail = (_lib.ptr(tap_gamma), _lib.ptr(tap_beta), _lib.ptr(tap), 0 if tap is None else tap.stride(0), M, E, H)
    if gact is not None:
        _call("ccd_proj_mlp_fused_gact", *head, _lib.ptr(gact), gact.stride(0), *tail)
    else:
        _call("ccd_proj_mlp_fused", *head, *tail)
    if span:
        span[1].record()
    kept = None if not save else ((xmid, y2, mean2, rstd2, u, gact) if gact is not None else (xmid, y2, mean2, rstd2, u))
    res = (out, yn, mean, rstd, kept)
    return res + (tap,) if tap is not None else res


def gemm_tn_colsum(a, b, out, colsum, *, splits=0):
    """out[P,Q] += a[Mc,P]^T @ b[Mc,Q] and colsum[P] += a.sum(0): the weight and the bias gradient of a Linear in one pass
    over dY (fp32 atomics)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(out, F32, "out"); _chk(colsum, F32, "colsum")
    Mc, Pd = a.shape
    Q = b.shape[1]
    assert b.shape[0] == Mc and tuple(out.shape) == (Pd, Q) and colsum.numel() == Pd
    span = TIMER.span("gemm_tn_atomic", 2.0 * Mc * Pd * Q, 2.0 * Mc * (Pd + Q) + 4.0 * Pd * Q) if TIMER is not None else None
    if span:
        span[0].record()
    _call("ccd_gemm_tn_colsum", _lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), Pd, Q, Mc, _lib.ptr(out), out.stride(0),
          _lib.ptr(colsum), int(splits))
    if span:
        span[1].record()
    return out


_TN_WS = {}
_TN_WS_RETIRED = []      # outgrown workspaces stay allocated: a captured HIP graph (pretrain.GraphedTrainingStep) may still write to them


def tn_pair_workspace(device, P1, Q1, P2, Q2):
    """The split-K workspace ccd_gemm_tn_pair_ws wants for these shapes: ONE fp32 buffer per (device, stream), grown to the largest
    request - launches on one stream use it in order, another stream gets its own.  A buffer that is outgrown is never freed (a graph
    captured earlier keeps its pointer).  None where the grouped kernel does not apply."""
    n = int(_lib.get().ccd_gemm_tn_pair_ws_floats(int(P1), int(Q1), int(P2), int(Q2)))
    if n <= 0:
        return None
    key = (device, _lib.stream())
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() < n:
        if ws is not None:
            _TN_WS_RETIRED.append(ws)
        ws = torch.empty(n, dtype=torch.float32, device=device)
        _TN_WS[key] = ws
    return ws


def gemm_tn_pair(a1, b1, out1, a2, b2, out2, *, workspace=True):
    """out1 += a1^T @ b1 and out2 += a2^T @ b2 (same number of contraction rows) in one launch where the shapes allow it
    (ccd_gemm_tn_pair_ws: per-slice partial tiles in a workspace + one reduction pass; workspace=False: ccd_gemm_tn_pair's fp32
    atomics); equal to two gemm_tn calls up to fp32 summation order."""
    for t, n in ((a1, "a1"), (b1, "b1"), (a2, "a2"), (b2, "b2")):
        _chk(t, BF16, n)
    _chk(out1, F32, "out1"); _chk(out2, F32, "out2")
    Mc = a1.shape[0]
    assert b1.shape[0] == Mc and a2.shape[0] == Mc and b2.shape[0] == Mc
    assert tuple(out1.shape) == (a1.shape[1], b1.shape[1]) and tuple(out2.shape) == (a2.shape[1], b2.shape[1])
    flops = 2.0 * Mc * (a1.shape[1] * b1.shape[1] + a2.shape[1] * b2.shape[1])
    nbytes = 2.0 * Mc * (a1.shape[1] + b1.shape[1] + a2.shape[1] + b2.shape[1]) + 4.0 * (out1.numel() + out2.numel())
    span = TIMER.span("gemm_tn_atomic", flops, nbytes) if TIMER is not None else None
    if span:
        span[0].record()
    ws = tn_pair_workspace(a1.device, a1.shape[1], b1.shape[1], a2.shape[1], b2.shape[1]) if workspace else None
    _call("ccd_gemm_tn_pair_ws", _lib.ptr(a1), a1.stride(0), _lib.ptr(b1), b1.stride(0), a1.shape[1], b1.shape[1], _lib.ptr(out1),
          out1.stride(0), _lib.ptr(a2), a2.stride(0), _lib.ptr(b2), b2.stride(0), a2.shape[1], b2.shape[1], _lib.ptr(out2),
          out2.stride(0), Mc, _lib.ptr(ws), ws.numel() if ws is not None else 0)
    if span:
        span[1].record()


def gemm_tn(a, b, out, *, accumulate=True, alpha=1.0, splits=0, d_rows=None, rows_mul=1):
    """out[P,Q] (+)= a[Mc,P]^T @ b[Mc,Q]  (fp32 out; accumulate=True adds with fp32 atomics, split over Mc)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(out, F32, "out")
    Mc, Pd = a.shape
    Q = b.shape[1]
    assert b.shape[0] == Mc and tuple(out.shape) == (Pd, Q)
    lib = _lib.get()
    span = TIMER.span("gemm_tn_" + ("atomic" if accumulate else "f32"), 2.0 * Mc * Pd * Q,
                      2.0 * Mc * (Pd + Q) + 4.0 * Pd * Q) \
        if TIMER is not None and d_rows is None else None
    if span:
        span[0].record()
    _lib.check(lib.ccd_gemm_tn(_lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), Pd, Q, Mc,
                               EPI_ATOMIC if accumulate else EPI_F32, _lib.ptr(out), out.stride(0), float(alpha),
                               int(splits) if accumulate else 1, _lib.ptr(d_rows), int(rows_mul), _lib.stream()),
               "gemm_tn")
    if span:
        span[1].record()
    return out


def ln_fwd(x, gamma, beta, eps=1e-6):
    """x [rows,E] fp32 -> (y bf16, mean, rstd)."""
    _chk(x, F32, "x")
    rows, E = x.shape
    y = torch.empty((rows, E), dtype=BF16, device=x.device)
    mean = torch.empty(rows, dtype=F32, device=x.device)
    rstd = torch.empty(rows, dtype=F32, device=x.device)
    with _Span("layernorm_fwd", 8.0 * rows * E, 6.0 * rows * E):
        _lib.check(_lib.get().ccd_ln_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), _lib.ptr(mean),
                                         _lib.ptr(rstd), rows, E, float(eps), _lib.stream()), "ln_fwd")
    return y, mean, rstd


def ln_bwd(dy, x, mean, rstd, gamma, g, dgamma, dbeta, accumulate=True, gb=None, rowscale=None, rows_per_sample=1,
           dbias=None):
    """g (+)= LN'(dy); dgamma += , dbeta += (in place).  Optional fused tail: gb = bf16(g * rowscale), dbias += colsum(gb)."""
    g16 = g.dtype == BF16
    _chk(dy, BF16, "dy"); _chk(x, F32, "x"); _chk(g, BF16 if g16 else F32, "g"); _chk(gb, BF16, "gb")
    rows, E = x.shape
    fn = _lib.get().ccd_ln_bwd_g16 if g16 else _lib.get().ccd_ln_bwd
    gbytes = 2.0 if g16 else 4.0
    with _Span("layernorm_bwd", 16.0 * rows * E, rows * E * (6.0 + (2 * gbytes if accumulate else gbytes) + (2.0 if gb is not None else 0.0))):
        _lib.check(fn(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma),
                                         _lib.ptr(g), 1 if accumulate else 0, _lib.ptr(dgamma), _lib.ptr(dbeta),
                                         _lib.ptr(gb), _lib.ptr(rowscale), int(rows_per_sample), _lib.ptr(dbias), rows, E,
                                         _lib.stream()), "ln_bwd")
    return g


def attention_fwd(qkv, heads, scale):
    """qkv [views,256,3*E] bf16 -> (out [views,256,E] bf16, lse [views,heads,256] fp32)."""
    _chk(qkv, BF16, "qkv")
    views, T, E3 = qkv.shape
    assert T == 256 and E3 == 3 * heads * 64 and qkv.is_contiguous()
    out = torch.empty((views, T, E3 // 3), dtype=BF16, device=qkv.device)
    lse = torch.empty((views, heads, T), dtype=F32, device=qkv.device)
    # per (view, head): S = Q K^T and O = P V, 2 * 256 * 256 * 64 flop each; q, k, v read and o written once
    with _Span("attention_fwd", views * heads * 4.0 * T * T * 64, views * heads * (4.0 * T * 64 * 2 + 4.0 * T)):
        _lib.check(_lib.get().ccd_attention_fwd(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse), views, heads, float(scale),
                                                _lib.stream()), "attention_fwd")
    return out, lse


def attention_bwd(qkv, out, d_out, lse, heads, scale, d_bias=None, dout_colsum=None, dout_colsum_mat=None):
    """-> d_qkv bf16 [views,256,3E].  d_bias (fp32 [3E], optional): += the qkv-bias gradient (column sums of d_qkv) without a
    pass over d_qkv - q part inside the dQ kernel, k part identically 0, v part = colsum(d_out) = `dout_colsum` [E], or
    `dout_colsum` [E] @ `dout_colsum_mat` [E, E] when the caller knows d_out = gb @ mat (see include/ccd_hip.h)."""
    _chk(qkv, BF16, "qkv"); _chk(out, BF16, "out"); _chk(d_out, BF16, "d_out"); _chk(d_bias, F32, "d_bias")
    _chk(dout_colsum, F32, "dout_colsum"); _chk(dout_colsum_mat, F32, "dout_colsum_mat")
    assert qkv.is_contiguous() and out.is_contiguous() and d_out.is_contiguous()
    views = qkv.shape[0]
    d_qkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    ws = None
    if d_bias is not None:
        assert d_bias.numel() == qkv.shape[2] and d_bias.is_contiguous() and dout_colsum is not None
        assert dout_colsum.numel() == qkv.shape[2] // 3 and dout_colsum.is_contiguous()
        ws = torch.empty(int(_lib.get().ccd_attention_bwd_ws_floats(views, heads)), dtype=F32, device=qkv.device)
    # five products (S, dP, dV, dK, dQ) of 2 * 256 * 256 * 64 flop per (view, head); q, k, v, o, dO read and dq, dk, dv written once
    with _Span("attention_bwd", views * heads * 10.0 * 256 * 256 * 64, views * heads * (8.0 * 256 * 64 * 2 + 8.0 * 256)):
        _lib.check(_lib.get().ccd_attention_bwd(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(d_out), _lib.ptr(lse),
                                                _lib.ptr(delta), _lib.ptr(d_qkv), views, heads, float(scale),
                                                _lib.ptr(d_bias), _lib.ptr(ws), _lib.ptr(dout_colsum), _lib.ptr(dout_colsum_mat),
                                                0 if dout_colsum_mat is None else dout_colsum_mat.stride(0), _lib.stream()),
                   "attention_bwd")
    return d_qkv


U8, I32 = torch.uint8, torch.int32


def _call(name, *args):
    _lib.check(getattr(_lib.get(), name)(*args, _lib.stream()), name)


# ------------------------------------------------------------------------------------------ patch embed & helpers
def patch_embed_fwd(img, w, bias, pos, out=None):
    views, E = img.shape[0], w.shape[0]
    assert img.dtype == F32 and img.is_contiguous() and tuple(img.shape[1:]) == (3, 32, 128)
    if out is None:
        out = torch.empty((views * 256, E), dtype=F32, device=img.device)
    _call("ccd_patch_embed_fwd", _lib.ptr(img), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(pos), _lib.ptr(out), views, E)
    return out


def patch_embed_bwd(img, g, d_w, d_bias, d_pos):
    views, E = img.shape[0], g.shape[-1]
    assert img.dtype == F32 and img.is_contiguous() and g.dtype in (F32, BF16) and g.is_contiguous()
    ws_p = torch.empty((views * 256, 48), dtype=BF16, device=g.device)
    if g.dtype == BF16:
        _call("ccd_patch_embed_bwd_g16", _lib.ptr(img), _lib.ptr(g), _lib.ptr(d_w), _lib.ptr(d_bias), _lib.ptr(d_pos), _lib.ptr(ws_p), views, E)
        return
    ws_g = torch.empty((views * 256, E), dtype=BF16, device=g.device)
    _call("ccd_patch_embed_bwd", _lib.ptr(img), _lib.ptr(g), _lib.ptr(d_w), _lib.ptr(d_bias), _lib.ptr(d_pos),
          _lib.ptr(ws_g), _lib.ptr(ws_p), views, E)


def small_matmul(a, b, out, trans_a=False, accumulate=False):
    """out[M,N] (+)= op(a) @ b, fp32."""
    K, N = b.shape
    M = a.shape[1] if trans_a else a.shape[0]
    _call("ccd_small_matmul_f32", _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), M, N, K, int(trans_a), int(accumulate))
    return out


def colsum_bf16(x, out, d_rows=None, rows_mul=1):
    rows, N = x.shape
    _call("ccd_colsum_bf16", _lib.ptr(x), x.stride(0), rows, N, _lib.ptr(d_rows), int(rows_mul), _lib.ptr(out))
    return out


def cast_bf16(src, dst):
    _call("ccd_cast_bf16", _lib.ptr(src), _lib.ptr(dst), src.numel())
    return dst


def scale_cast_rows(src, dst, rowscale=None, rows_per_sample=1):
    rows, E = src.shape
    _call("ccd_scale_cast_rows", _lib.ptr(src), _lib.ptr(dst), _lib.ptr(rowscale), int(rows_per_sample), rows, E)
    return dst


def mirror_bf16(descs_dev, ndesc, total_tiles):
    _call("ccd_mirror_bf16", _lib.ptr(descs_dev), ndesc, total_tiles)


# ------------------------------------------------------------------------------------------ character-region path
def ccl_label(mask):
    """mask [B,32,128] fp32 (nonzero = text) -> uint8 id map [B,32,128] (255 = background)."""
    assert mask.dtype == F32 and mask.is_contiguous() and tuple(mask.shape[1:]) == (32, 128)
    out = torch.empty(mask.shape, dtype=U8, device=mask.device)
    _call("ccd_ccl_label", _lib.ptr(mask), _lib.ptr(out), mask.shape[0])
    return out


def mask_to_idmap(mask):
    out = torch.empty(mask.shape, dtype=U8, device=mask.device)
    _call("ccd_mask_to_idmap", _lib.ptr(mask), _lib.ptr(out), mask.shape[0])
    return out


def seg_to_mask(seg_logits, images):
    out = torch.empty((images, 32, 128), dtype=F32, device=seg_logits.device)
    _call("ccd_seg_to_mask", _lib.ptr(seg_logits), _lib.ptr(out), images)
    return out


def kmeans2_mask(grays, device=None):
    """Text masks of word images (clusterpixels(im, 2), mask_create/generate_mask.py:13-29) for a list of uint8 [h, w] gray
    images of any sizes -> list of uint8 [h, w] 0/1 arrays (numpy).  One workgroup per image of the ragged batch."""
    import numpy as np
    if not grays:
        return []
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    arrs = [np.ascontiguousarray(np.asarray(g, dtype=np.uint8)) for g in grays]
    for a in arrs:
        assert a.ndim == 2 and a.size > 0, "kmeans2_mask expects non-empty [h, w] gray images"
    sizes = np.array([a.size for a in arrs], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    hw = np.array([a.shape for a in arrs], dtype=np.int32).reshape(-1)
    flat = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(dev)
    d_offs, d_hw = torch.from_numpy(offs).to(dev), torch.from_numpy(hw).to(dev)
    out = torch.empty_like(flat)
    _call("ccd_kmeans2_mask", _lib.ptr(flat), _lib.ptr(d_offs), _lib.ptr(d_hw), _lib.ptr(out), len(arrs))
    host = out.cpu().numpy()
    return [host[offs[i]:offs[i + 1]].reshape(arrs[i].shape) for i in range(len(arrs))]


def augment_views(img, params, theta, mean, std, overlay=None, warp_maps=None):
    """img uint8 [B,H,W,3] (resized samples), params fp32 [B,2,96], theta fp32 [B,3,3] -> image_tensors fp32 [B,3,3,H,W]:
    (plain, colour-augmented, colour-augmented + warped by theta), normalised - the dataset's batch contract
    (datasetsupervised_kmeans.py:48-87).  The neighbourhood members (JPEG, blurs, convolutions) run in a pre-pass that stages
    one uint8 image per (sample, view).  overlay: fp16 [layers, 2, H, W] (alpha, intensity) cloud layers that rows with a `weather`
    member refer to (ccd_amd/dataset/weather.py), blended at the end of the pre-pass.  warp_maps: fp32 [maps, 2, H, W] source
    positions of the piecewise-affine warps that view-2 rows with params[84] = m > 0 take instead of theta (map m - 1)."""
    import ctypes as C
    assert img.dtype == U8 and img.is_contiguous() and img.dim() == 4 and img.shape[3] == 3
    _chk(params, F32, "params"); _chk(theta, F32, "theta")
    B, H, W, _ = img.shape
    assert tuple(params.shape) == (B, 2, 96) and tuple(theta.shape) == (B, 3, 3) and params.is_contiguous()
    out = torch.empty((B, 3, 3, H, W), dtype=F32, device=img.device)
    staged = torch.empty((B, 2, H, W, 3), dtype=U8, device=img.device)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    layers = 0
    if overlay is not None:
        assert overlay.dtype == torch.float16 and overlay.is_contiguous() and tuple(overlay.shape[1:]) == (2, H, W)
        layers = overlay.shape[0]
    maps = 0
    if warp_maps is not None:
        assert warp_maps.dtype == F32 and warp_maps.is_contiguous() and tuple(warp_maps.shape[1:]) == (2, H, W)
        maps = warp_maps.shape[0]
    _call("ccd_augment_views", _lib.ptr(img), _lib.ptr(params), _lib.ptr(theta), _lib.ptr(out), _lib.ptr(staged), B, H, W,
          C.cast(m3, C.c_void_p), C.cast(s3, C.c_void_p), _lib.ptr(overlay), int(layers), _lib.ptr(warp_maps), int(maps))
    return out


def warp_idmap(src, theta):
    """src uint8 [B,32,128], theta fp32 [B,3,3] (or [B,2,3]) -> warped id map (view 2)."""
    assert src.dtype == U8 and src.is_contiguous() and theta.dtype == F32 and theta.is_contiguous()
    out = torch.empty_like(src)
    _call("ccd_warp_idmap", _lib.ptr(src), _lib.ptr(theta), theta.stride(0), _lib.ptr(out), src.shape[0])
    return out


def region_stats(idmap):
    views = idmap.shape[0]
    dev = idmap.device
    tok_plane = torch.empty((views, 256, 4), dtype=U8, device=dev)      # up to 4 (plane, coefficient) pairs per token
    tok_coef = torch.empty((views, 256, 4), dtype=F32, device=dev)
    present = torch.empty((views, 26), dtype=U8, device=dev)
    _call("ccd_region_stats", _lib.ptr(idmap), _lib.ptr(tok_plane), _lib.ptr(tok_coef), _lib.ptr(present), views)
    return tok_plane, tok_coef, present


def select_scan(present, batch):
    dev = present.device
    nsel = torch.empty(batch, dtype=I32, device=dev)
    offset = torch.empty(batch, dtype=I32, device=dev)
    total = torch.empty(1, dtype=I32, device=dev)
    new_index = torch.empty((batch, 26), dtype=U8, device=dev)
    _call("ccd_select_scan", _lib.ptr(present), batch, _lib.ptr(nsel), _lib.ptr(offset), _lib.ptr(total),
          _lib.ptr(new_index))
    return nsel, offset, total, new_index


def region_pool_fwd(feat, tok_plane, tok_coef, nsel, offset, total, rows, batch):
    E = feat.shape[-1]
    _call("ccd_region_pool_fwd", _lib.ptr(feat), _lib.ptr(tok_plane), _lib.ptr(tok_coef), _lib.ptr(nsel),
          _lib.ptr(offset), _lib.ptr(total), _lib.ptr(rows), batch, E)
    return rows


def region_pool_bwd(d_rows, tok_plane, tok_coef, nsel, offset, total, d_feat, batch):
    E = d_feat.shape[-1]
    _call("ccd_region_pool_bwd", _lib.ptr(d_rows), _lib.ptr(tok_plane), _lib.ptr(tok_coef), _lib.ptr(nsel),
          _lib.ptr(offset), _lib.ptr(total), _lib.ptr(d_feat), batch, E)
    return d_feat


def idmap_to_planes(idmap):
    out = torch.empty((idmap.shape[0], 26, 32, 128), dtype=F32, device=idmap.device)
    _call("ccd_idmap_to_planes", _lib.ptr(idmap), _lib.ptr(out), idmap.shape[0])
    return out


def planes_to_idmap(planes):
    planes = planes.contiguous().float()
    out = torch.empty((planes.shape[0], 32, 128), dtype=U8, device=planes.device)
    _call("ccd_planes_to_idmap", _lib.ptr(planes), _lib.ptr(out), planes.shape[0])
    return out


# ------------------------------------------------------------------------------------------ DINO head pieces
def l2norm_fwd(x, y, inv, d_rows=None, rows_mul=1):
    _call("ccd_l2norm_fwd", _lib.ptr(x), _lib.ptr(y), _lib.ptr(inv), x.shape[0], _lib.ptr(d_rows), rows_mul, x.shape[1])


def l2norm_bwd(x, inv, dy, dx, d_rows=None, rows_mul=1):
    _call("ccd_l2norm_bwd", _lib.ptr(x), _lib.ptr(inv), _lib.ptr(dy), _lib.ptr(dx), x.shape[0], _lib.ptr(d_rows),
          rows_mul, x.shape[1])


def weightnorm_fwd(v, g, w, w_t, inv):
    K, D = v.shape
    _call("ccd_weightnorm_fwd", _lib.ptr(v), _lib.ptr(g), _lib.ptr(w), _lib.ptr(w_t), _lib.ptr(inv), K, D)


def weightnorm_bwd(v, g, inv, dw, dv, dg):
    K, D = v.shape
    _call("ccd_weightnorm_bwd", _lib.ptr(v), _lib.ptr(g), _lib.ptr(inv), _lib.ptr(dw), _lib.ptr(dv), _lib.ptr(dg), K, D)


# ------------------------------------------------------------------------------------------ losses
def dino_loss_fwd(s_logits, t_logits, center, d_m, student_temp, teacher_temp, stats, loss_out):
    max_rows, K = s_logits.shape
    # (the kernels read the device-side row count M <= max_rows / 2; bytes are the worst case the launch is sized for)
    with _Span("dino_loss_fwd", 12.0 * max_rows * K, 8.0 * max_rows * K):
        _call("ccd_dino_loss_fwd", _lib.ptr(s_logits), _lib.ptr(t_logits), _lib.ptr(center), K, _lib.ptr(d_m), max_rows,
              float(student_temp), float(teacher_temp), _lib.ptr(stats), _lib.ptr(loss_out))


def dino_loss_bwd(s_logits, t_logits, center, d_m, student_temp, teacher_temp, stats, grad_scale, d_logits,
                  d_grad_scale=None):
    max_rows, K = s_logits.shape
    with _Span("dino_loss_bwd", 12.0 * max_rows * K, 10.0 * max_rows * K):
        _call("ccd_dino_loss_bwd", _lib.ptr(s_logits), _lib.ptr(t_logits), _lib.ptr(center), K, _lib.ptr(d_m), max_rows,
              float(student_temp), float(teacher_temp), _lib.ptr(stats), float(grad_scale), _lib.ptr(d_grad_scale),
              _lib.ptr(d_logits))


def head_loss_supported(K, D, max_rows):
    """True where ccd_head_loss_fwd / _bwd take the shape (D == 256, K % 512 == 0): the logits need not be materialised."""
    return D == 256 and K % 512 == 0 and max_rows > 0 and _lib.get().ccd_head_loss_ws_floats(int(max_rows), int(K)) > 0


_HEAD_LOSS_WS = {}


def _head_loss_ws(device, n):
    """One workspace per (device, stream) for the per-split partials of ccd_head_loss_fwd (grown, never shrunk)."""
    key = (device.index if device.type == "cuda" else -1, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _HEAD_LOSS_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _HEAD_LOSS_WS[key] = torch.empty(int(n), dtype=F32, device=device)
    return ws


def head_loss_fwd(zs, zt, ws, wt, center, d_m, student_temp, teacher_temp, stats, loss_out):
    """loss_out += DINO distillation loss of logits zs @ ws^T (student) against zt @ wt^T (teacher, centred) - the logits stay in
    registers (include/ccd_hip.h: ccd_head_loss_fwd).  stats [max_rows, 4] is what head_loss_bwd reads."""
    for t, n in ((zs, "zs"), (zt, "zt"), (ws, "ws"), (wt, "wt")):
        _chk(t, BF16, n)
    _chk(center, F32, "center"); _chk(stats, F32, "stats")
    max_rows, D = zs.shape
    K = ws.shape[0]
    assert zt.shape == zs.shape and wt.shape == ws.shape and ws.shape[1] == D and stats.shape == (max_rows, 4)
    part = _head_loss_ws(zs.device, _lib.get().ccd_head_loss_ws_floats(max_rows, K))
    with _Span("head_loss_fwd", 0.0, 4.0 * K * D):          # (the live row count 2M is device-side: no flop figure)
        _call("ccd_head_loss_fwd", _lib.ptr(zs), zs.stride(0), _lib.ptr(zt), zt.stride(0), _lib.ptr(ws), ws.stride(0), _lib.ptr(wt),
              wt.stride(0), _lib.ptr(center), K, D, _lib.ptr(d_m), max_rows, float(student_temp), float(teacher_temp), _lib.ptr(part),
              _lib.ptr(stats), _lib.ptr(loss_out))


def head_loss_bwd(zs, zt, ws, wt, center, d_m, student_temp, teacher_temp, stats, grad_scale, d_logits, d_grad_scale=None):
    """d_logits (bf16 [max_rows, K], rows < 2M written) = d loss / d (zs @ ws^T), the products recomputed (ccd_head_loss_bwd)."""
    _chk(d_logits, BF16, "d_logits")
    max_rows, D = zs.shape
    K = ws.shape[0]
    assert d_logits.shape == (max_rows, K)
    with _Span("head_loss_bwd", 0.0, 4.0 * K * D):
        _call("ccd_head_loss_bwd", _lib.ptr(zs), zs.stride(0), _lib.ptr(zt), zt.stride(0), _lib.ptr(ws), ws.stride(0), _lib.ptr(wt),
              wt.stride(0), _lib.ptr(center), K, D, _lib.ptr(d_m), max_rows, float(student_temp), float(teacher_temp), _lib.ptr(stats),
              float(grad_scale), _lib.ptr(d_grad_scale), _lib.ptr(d_logits), d_logits.stride(0))


def colsum_f32(x, out, d_rows=None, rows_mul=1):
    max_rows, K = x.shape
    _call("ccd_colsum_f32", _lib.ptr(x), K, _lib.ptr(d_rows), rows_mul, max_rows, _lib.ptr(out))


def matvec_bf16(w, v, out):
    """out[k] += w[k, :] . v   (w [K, D] bf16, v / out fp32; D % 256 == 0)."""
    _chk(w, BF16, "w"); _chk(v, F32, "v"); _chk(out, F32, "out")
    K, D = w.shape
    assert v.numel() == D and out.numel() == K
    _call("ccd_matvec_bf16", _lib.ptr(w), w.stride(0), _lib.ptr(v), K, D, _lib.ptr(out))
    return out


def center_ema(center, batch_sum, d_m, world, momentum):
    _call("ccd_center_ema", _lib.ptr(center), _lib.ptr(batch_sum), center.numel(), _lib.ptr(d_m), int(world),
          float(momentum))


def seg_loss(logits, mask_a, idmap_b, grad_scale, loss_out, d_logits=None):
    half = mask_a.shape[0]
    assert logits.shape[0] == 2 * half and logits.is_contiguous()
    _call("ccd_seg_loss", _lib.ptr(logits), _lib.ptr(mask_a), _lib.ptr(idmap_b), half, float(grad_scale),
          _lib.ptr(loss_out), _lib.ptr(d_logits))


# ------------------------------------------------------------------------------------------ optimiser
def seg_sumsq(grad, chunk_seg, chunk_begin, chunk_len, norm2):
    _call("ccd_seg_sumsq", _lib.ptr(grad), _lib.ptr(chunk_seg), _lib.ptr(chunk_begin), _lib.ptr(chunk_len),
          chunk_seg.numel(), _lib.ptr(norm2))


def adamw(param, grad, exp_avg, exp_avg_sq, mirror, chunk_seg, chunk_begin, chunk_len, hyper, norm2, clip,
          beta1=0.9, beta2=0.999, eps=1e-8):
    _call("ccd_adamw", _lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), _lib.ptr(mirror),
          _lib.ptr(chunk_seg), _lib.ptr(chunk_begin), _lib.ptr(chunk_len), chunk_seg.numel(), _lib.ptr(hyper),
          _lib.ptr(norm2), float(clip), float(beta1), float(beta2), float(eps))


def clip_scale(grad, chunk_seg, chunk_begin, chunk_len, norm2, clip):
    _call("ccd_clip_scale", _lib.ptr(grad), _lib.ptr(chunk_seg), _lib.ptr(chunk_begin), _lib.ptr(chunk_len),
          chunk_seg.numel(), _lib.ptr(norm2), float(clip))


def ema(teacher, student, mirror, m, d_m=None):
    """teacher = m teacher + (1 - m) student (+ bf16 mirror); d_m (fp32 [2] on the device: {m, 1 - m}) overrides m at run time."""
    _call("ccd_ema", _lib.ptr(teacher), _lib.ptr(student), _lib.ptr(mirror), teacher.numel(), float(m), float(1.0 - m), _lib.ptr(d_m))


# ------------------------------------------------------------------------------------------ segmentation head
class ConvDesc(ctypes.Structure):
    """ccd_conv_desc of include/ccd_hip.h (host-side, passed by pointer)."""
    _fields_ = [("g_h_log2", ctypes.c_int), ("g_w_log2", ctypes.c_int), ("s_h", ctypes.c_int), ("s_w", ctypes.c_int),
                ("s_mul", ctypes.c_int), ("cin", ctypes.c_int), ("ntaps", ctypes.c_int),
                ("dy", ctypes.c_byte * 16), ("dx", ctypes.c_byte * 16),
                ("c_map", ctypes.c_int), ("c_py", ctypes.c_int), ("c_px", ctypes.c_int)]


def conv_desc(grid_hw, src_hw, cin, taps, s_mul=1, parity=None):
    """grid_hw: output-row grid (powers of two); taps: [(dy, dx)]; parity (py, px) turns on the 2x scatter."""
    gh, gw = grid_hw
    assert gh & (gh - 1) == 0 and gw & (gw - 1) == 0 and 1 <= len(taps) <= 16
    d = ConvDesc()
    d.g_h_log2, d.g_w_log2 = gh.bit_length() - 1, gw.bit_length() - 1
    d.s_h, d.s_w, d.s_mul, d.cin, d.ntaps = src_hw[0], src_hw[1], s_mul, cin, len(taps)
    for i, (a, b) in enumerate(taps):
        d.dy[i], d.dx[i] = a, b
    if parity is not None:
        d.c_map, d.c_py, d.c_px = 1, parity[0], parity[1]
    return d


def conv_gemm(src, desc, w, rows, out, *, bias=None, colsum=None, colsumsq=None):
    """out[rows -> c_map, N] (bf16) = gather(src)[rows, ntaps*cin] @ w[N, ntaps*cin]^T (+ bias); stats += column sums."""
    _chk(src, BF16, "src"); _chk(w, BF16, "w"); _chk(out, BF16, "out"); _chk(bias, F32, "bias")
    _chk(colsum, F32, "colsum"); _chk(colsumsq, F32, "colsumsq")
    N = w.shape[0]
    assert w.shape[1] == desc.ntaps * desc.cin and src.dim() == 2 and out.dim() == 2 and out.shape[1] >= N
    span = TIMER.span("conv_gemm", 2.0 * rows * N * w.shape[1],
                      2.0 * (src.numel() + w.numel() + rows * N)) if TIMER is not None else None
    if span:
        span[0].record()
    _call("ccd_conv_gemm", _lib.ptr(src), src.stride(0), ctypes.addressof(desc), _lib.ptr(w), w.stride(0), rows, N,
          _lib.ptr(out), out.stride(0), _lib.ptr(bias), _lib.ptr(colsum), _lib.ptr(colsumsq))
    if span:
        span[1].record()
    return out


def conv_wgrad(a, src, desc, out):
    """out[P, ntaps*cin] (fp32) += a[rows, P]^T @ gather(src)[rows, ntaps*cin] (implicit patch matrix)."""
    _chk(a, BF16, "a"); _chk(src, BF16, "src"); _chk(out, F32, "out")
    rows, Pd = a.shape
    assert tuple(out.shape) == (Pd, desc.ntaps * desc.cin)
    span = TIMER.span("conv_wgrad", 2.0 * rows * Pd * out.shape[1],
                      2.0 * (a.numel() + src.numel()) + 4.0 * out.numel()) if TIMER is not None else None
    if span:
        span[0].record()
    _call("ccd_conv_wgrad", _lib.ptr(a), a.stride(0), Pd, _lib.ptr(src), src.stride(0), ctypes.addressof(desc), rows,
          _lib.ptr(out), out.stride(0))
    if span:
        span[1].record()
    return out


def im2col(src, desc, rows, out=None):
    _chk(src, BF16, "src")
    if out is None:
        out = torch.empty((rows, desc.ntaps * desc.cin), dtype=BF16, device=src.device)
    assert out.is_contiguous() and out.numel() == rows * desc.ntaps * desc.cin
    _call("ccd_im2col", _lib.ptr(src), src.stride(0), ctypes.addressof(desc), rows, _lib.ptr(out))
    return out


def bn_finalize(stats, count, eps, momentum, mean_rstd, running_mean, running_var):
    C = running_mean.numel()
    assert stats.numel() == 2 * C and mean_rstd.numel() == 2 * C
    _call("ccd_bn_finalize", _lib.ptr(stats), float(count), float(eps), float(momentum), _lib.ptr(mean_rstd),
          _lib.ptr(running_mean), _lib.ptr(running_var), C)


def bn_relu_fwd(x, mean_rstd, gamma, beta, out):
    _chk(x, BF16, "x"); _chk(out, BF16, "out")
    rows, C = x.shape
    _call("ccd_bn_relu_fwd", _lib.ptr(x), x.stride(0), _lib.ptr(mean_rstd), _lib.ptr(gamma), _lib.ptr(beta),
          _lib.ptr(out), out.stride(0), rows, C)
    return out


def bn_relu_bwd_reduce(dy, x, mean_rstd, gamma, beta, red):
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(red, F32, "red")
    rows, C = x.shape
    _call("ccd_bn_relu_bwd_reduce", _lib.ptr(dy), dy.stride(0), _lib.ptr(x), x.stride(0), _lib.ptr(mean_rstd),
          _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(red), rows, C)


def bn_relu_bwd_apply(dy, x, mean_rstd, gamma, beta, red, count, red_local, dgamma, dbeta, dx):
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(dx, BF16, "dx")
    rows, C = x.shape
    _call("ccd_bn_relu_bwd_apply", _lib.ptr(dy), dy.stride(0), _lib.ptr(x), x.stride(0), _lib.ptr(mean_rstd),
          _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(red), float(count), _lib.ptr(red_local), _lib.ptr(dgamma),
          _lib.ptr(dbeta), _lib.ptr(dx), dx.stride(0), rows, C)
    return dx


def cls_gather_fwd(zT, bias, images, H, W):
    """zT fp32 [>=18, pixels] (row co*9+tap) -> fp32 logits [images, 2, H, W]."""
    _chk(zT, F32, "zT"); _chk(bias, F32, "bias")
    assert zT.shape[0] >= 18 and zT.shape[1] == images * H * W
    logits = torch.empty((images, 2, H, W), dtype=F32, device=zT.device)
    _call("ccd_cls_gather_fwd", _lib.ptr(zT), zT.stride(0), _lib.ptr(bias), _lib.ptr(logits), images, H, W)
    return logits


def cls_grad_cols(dlogits, images, H, W):
    """fp32 dlogits [images, 2, H, W] -> bf16 g [pixels, 64] (column co*9+tap = shifted gradient plane)."""
    _chk(dlogits, F32, "dlogits")
    assert dlogits.is_contiguous() and tuple(dlogits.shape) == (images, 2, H, W)
    g = torch.empty((images * H * W, 64), dtype=BF16, device=dlogits.device)
    _call("ccd_cls_grad_cols", _lib.ptr(dlogits), _lib.ptr(g), images, H, W)
    return g


def cls_tail_supported(y, H, W):
    """The fused BatchNorm + ReLU + classifier kernels exist for the reference's head shape only (kernels/cls_tail.h)."""
    return y.shape[1] == 128 and (H, W) == (32, 128) and y.stride(0) % 8 == 0


def cls_tail_fwd(y, mean_rstd, gamma, beta, w, bias, images, H, W):
    """logits fp32 [images, 2, H, W] = Conv2d(C, 2, 3, padding=1)(relu(bn(y))); y bf16 [images*H*W, C] BEFORE its BatchNorm."""
    _chk(y, BF16, "y"); _chk(mean_rstd, F32, "mean_rstd"); _chk(w, F32, "w"); _chk(bias, F32, "bias")
    C = y.shape[1]
    assert y.shape[0] == images * H * W and y.stride(1) == 1 and w.is_contiguous() and tuple(w.shape) == (2, C, 3, 3)
    logits = torch.empty((images, 2, H, W), dtype=F32, device=y.device)
    _call("ccd_cls_tail_fwd", _lib.ptr(y), y.stride(0), _lib.ptr(mean_rstd), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(w),
          _lib.ptr(bias), _lib.ptr(logits), images, H, W, C)
    return logits


def cls_tail_bwd_reduce(dlogits, y, mean_rstd, gamma, beta, w, red, db_cls, images, H, W):
    """red [2C] += BatchNorm's two backward sums of d(relu(bn(y))) under the classifier; db_cls [2] += sum dlogits."""
    _chk(dlogits, F32, "dlogits"); _chk(y, BF16, "y"); _chk(red, F32, "red"); _chk(db_cls, F32, "db_cls")
    C = y.shape[1]
    assert dlogits.is_contiguous() and tuple(dlogits.shape) == (images, 2, H, W) and y.shape[0] == images * H * W
    assert w.is_contiguous() and red.numel() == 2 * C and red.is_contiguous() and db_cls.is_contiguous()
    _call("ccd_cls_tail_bwd_reduce", _lib.ptr(dlogits), _lib.ptr(y), y.stride(0), _lib.ptr(mean_rstd), _lib.ptr(gamma),
          _lib.ptr(beta), _lib.ptr(w), _lib.ptr(red), _lib.ptr(db_cls), images, H, W, C)
    return red


def cls_tail_bwd_apply(dlogits, y, mean_rstd, gamma, beta, w, red, count, red_local, dgamma, dbeta, dw_cls, dbias_t, dy,
                       images, H, W):
    """dy bf16 [images*H*W, C] = gradient w.r.t. y; dgamma / dbeta / dw_cls [2, C, 3, 3] / dbias_t [C] accumulate (fp32)."""
    _chk(dlogits, F32, "dlogits"); _chk(y, BF16, "y"); _chk(dy, BF16, "dy"); _chk(dw_cls, F32, "dw_cls")
    C = y.shape[1]
    assert dlogits.is_contiguous() and y.shape[0] == images * H * W and dy.shape == y.shape and dy.stride(1) == 1
    assert dw_cls.is_contiguous() and tuple(dw_cls.shape) == (2, C, 3, 3) and dbias_t.is_contiguous() and dbias_t.numel() == C
    assert dgamma.is_contiguous() and dbeta.is_contiguous() and red.is_contiguous() and red_local.is_contiguous()
    _call("ccd_cls_tail_bwd_apply", _lib.ptr(dlogits), _lib.ptr(y), y.stride(0), _lib.ptr(mean_rstd), _lib.ptr(gamma),
          _lib.ptr(beta), _lib.ptr(w), _lib.ptr(red), float(count), _lib.ptr(red_local), _lib.ptr(dgamma), _lib.ptr(dbeta),
          _lib.ptr(dw_cls), _lib.ptr(dbias_t), _lib.ptr(dy), dy.stride(0), images, H, W, C)
    return dy


def permute4(src, strides, dims, dst, accumulate=False, dst_strides=None):
    """dst[idx . dst_strides] (bf16 cast, or fp32 += when accumulate) <- src.flatten()[idx . strides]; dst_strides
    default to contiguous over `dims`."""
    _chk(src, F32, "src"); _chk(dst, F32 if accumulate else BF16, "dst")
    n = list(dims) + [1] * (4 - len(dims))
    s = list(strides) + [0] * (4 - len(strides))
    if dst_strides is None:
        d = [n[1] * n[2] * n[3], n[2] * n[3], n[3], 1]
        assert dst.is_contiguous() and dst.numel() == n[0] * n[1] * n[2] * n[3]
    else:
        d = list(dst_strides) + [0] * (4 - len(dst_strides))
    arr_l, arr_i = ctypes.c_long * 4, ctypes.c_int * 4
    _call("ccd_permute4", _lib.ptr(src), arr_l(*s), arr_l(*d), arr_i(*n), _lib.ptr(dst), 1 if accumulate else 0)
    return dst


class _PermuteJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("src_strides", ctypes.c_long * 4),
                ("dst_strides", ctypes.c_long * 4), ("dims", ctypes.c_int * 4)]


PERMUTE_MULTI_MAX = 24


def permute4_multi(jobs, accumulate=False):
    """jobs: [(src, strides, dims, dst[, dst_strides])] - ops.permute4's arguments, up to 24 of them in ONE launch."""
    for lo in range(0, len(jobs), PERMUTE_MULTI_MAX):
        part = jobs[lo:lo + PERMUTE_MULTI_MAX]
        arr = (_PermuteJob * len(part))()
        for k, job in enumerate(part):
            src, strides, dims, dst = job[:4]
            dst_strides = job[4] if len(job) > 4 else None
            _chk(src, F32, "src"); _chk(dst, F32 if accumulate else BF16, "dst")
            n = list(dims) + [1] * (4 - len(dims))
            s_ = list(strides) + [0] * (4 - len(strides))
            if dst_strides is None:
                d = [n[1] * n[2] * n[3], n[2] * n[3], n[3], 1]
                assert dst.is_contiguous() and dst.numel() == n[0] * n[1] * n[2] * n[3]
            else:
                d = list(dst_strides) + [0] * (4 - len(dst_strides))
            arr[k].src, arr[k].dst = _lib.ptr(src), _lib.ptr(dst)
            arr[k].src_strides[:], arr[k].dst_strides[:], arr[k].dims[:] = s_, d, n
        _call("ccd_permute4_multi", ctypes.cast(arr, ctypes.c_void_p).value, len(part), 1 if accumulate else 0)


class _BnFinalizeJob(ctypes.Structure):
    _fields_ = [("stats", ctypes.c_void_p), ("mean_rstd", ctypes.c_void_p), ("running_mean", ctypes.c_void_p),
                ("running_var", ctypes.c_void_p), ("batches", ctypes.c_void_p), ("count", ctypes.c_float), ("eps", ctypes.c_float),
                ("momentum", ctypes.c_float), ("C", ctypes.c_int)]


def bn_finalize_multi(jobs):
    """jobs: [(stats, count, eps, momentum, mean_rstd, running_mean, running_var, num_batches_tracked or None)], at most 4: the
    BatchNorm layers of one level in one launch (mean / rstd, running statistics, batch counter)."""
    assert 1 <= len(jobs) <= 4
    arr = (_BnFinalizeJob * len(jobs))()
    for k, (stats, count, eps, momentum, mean_rstd, rm, rv, nb) in enumerate(jobs):
        C = rm.numel()
        assert stats.numel() == 2 * C and mean_rstd.numel() == 2 * C and (nb is None or nb.dtype == torch.int64)
        arr[k].stats, arr[k].mean_rstd, arr[k].running_mean, arr[k].running_var = (_lib.ptr(stats), _lib.ptr(mean_rstd), _lib.ptr(rm),
                                                                                    _lib.ptr(rv))
        arr[k].batches = _lib.ptr(nb) if nb is not None else None
        arr[k].count, arr[k].eps, arr[k].momentum, arr[k].C = float(count), float(eps), float(momentum), C
    _call("ccd_bn_finalize_multi", ctypes.cast(arr, ctypes.c_void_p).value, len(jobs))


# ------------------------------------------------------------------------------------------------ finetune path
I64 = torch.int64


def dropout(src, p, seed, *, resid=None, out=None, out_dtype=None):
    """out = (resid if given) + Dropout_p(src) with the counter-based mask of `seed` (ccd_hip.h: ccd_dropout)."""
    assert src.is_contiguous() and src.dtype in (F32, BF16)
    if out is None:
        out = torch.empty(src.shape, dtype=out_dtype or src.dtype, device=src.device)
    assert out.is_contiguous() and out.shape == src.shape and out.dtype in (F32, BF16)
    _chk(resid, F32, "resid")
    _call("ccd_dropout", _lib.ptr(src), int(src.dtype == BF16), _lib.ptr(resid), _lib.ptr(out), int(out.dtype == BF16),
          src.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, float(p))
    return out


def droppath_scales(keep, samples, seed, d_seed=None):
    """keep fp32 [depth] (device) -> fp32 [depth, 2, samples]: per-(block, branch, sample) DropPath scale (0 or 1/keep).
    d_seed (int64 [1] on the device, optional) is added to `seed` when the kernel runs (HIP-graph replays)."""
    _chk(keep, F32, "keep")
    out = torch.empty((keep.shape[0], 2, samples), dtype=F32, device=keep.device)
    _call("ccd_droppath_scales", _lib.ptr(keep), _lib.ptr(out), 2 * samples, keep.shape[0], int(seed) & 0xFFFFFFFFFFFFFFFF,
          _lib.ptr(d_seed))
    return out


def dec_embed_fwd(tokens, emb, pos, p=0.0, seed=0):
    """tokens int64 [B,T] -> x fp32 [B*T, D] = dropout(emb[tokens] + pos[:T])."""
    assert tokens.dtype == I64 and tokens.is_contiguous()
    _chk(emb, F32, "emb"); _chk(pos, F32, "pos")
    B, T = tokens.shape
    D = emb.shape[1]
    x = torch.empty((B * T, D), dtype=F32, device=emb.device)
    _call("ccd_dec_embed_fwd", _lib.ptr(tokens), _lib.ptr(emb), _lib.ptr(pos), _lib.ptr(x), B * T, T, D, emb.shape[0],
          int(seed) & 0xFFFFFFFFFFFFFFFF, float(p))
    return x


def dec_embed_bwd(tokens, dx, demb, padding_idx, p=0.0, seed=0):
    _chk(dx, F32, "dx"); _chk(demb, F32, "demb")
    _call("ccd_dec_embed_bwd", _lib.ptr(tokens), _lib.ptr(dx), _lib.ptr(demb), tokens.numel(), dx.shape[1], demb.shape[0],
          int(padding_idx), int(seed) & 0xFFFFFFFFFFFFFFFF, float(p))


def dec_attn_fwd(q, k, v, B, H, Tq, Tk, scale, *, tokens=None, key_len=None, pad_idx=-1, causal=False, p=0.0, seed=0,
                 want_probs=False):
    """q [B*Tq, >=64H] / k, v [B*Tk, >=64H] bf16 2-D views (row stride = .stride(0)) -> (out bf16 [B*Tq, 64H], lse, probs)."""
    for t_ in (q, k, v):
        assert t_.dtype == BF16 and t_.dim() == 2 and t_.stride(1) == 1
    out = torch.empty((B * Tq, 64 * H), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, Tq), dtype=F32, device=q.device)
    probs = torch.empty((B, H, Tq, Tk), dtype=F32, device=q.device) if want_probs else None
    _call("ccd_dec_attn_fwd", _lib.ptr(q), q.stride(0), _lib.ptr(k), k.stride(0), _lib.ptr(v), v.stride(0), _lib.ptr(out),
          out.stride(0), _lib.ptr(lse), _lib.ptr(probs), _lib.ptr(tokens), _lib.ptr(key_len), int(pad_idx), int(causal),
          B, H, Tq, Tk, float(scale), int(seed) & 0xFFFFFFFFFFFFFFFF, float(p))
    return out, lse, probs


def dec_attn_bwd(q, k, v, out, d_out, lse, dq, dk, dv, B, H, Tq, Tk, scale, *, tokens=None, key_len=None, pad_idx=-1,
                 causal=False, p=0.0, seed=0):
    """dq / dk / dv: preallocated bf16 2-D views the gradients are written into (all rows, head columns only)."""
    for t_ in (q, k, v, dq, dk, dv):
        assert t_.dtype == BF16 and t_.dim() == 2 and t_.stride(1) == 1
    _chk(out, BF16, "out"); _chk(d_out, BF16, "d_out")
    assert out.stride(0) == d_out.stride(0)
    _call("ccd_dec_attn_bwd", _lib.ptr(q), q.stride(0), _lib.ptr(k), k.stride(0), _lib.ptr(v), v.stride(0), _lib.ptr(out),
          _lib.ptr(d_out), out.stride(0), _lib.ptr(lse), _lib.ptr(tokens), _lib.ptr(key_len), int(pad_idx), int(causal),
          B, H, Tq, Tk, float(scale), int(seed) & 0xFFFFFFFFFFFFFFFF, float(p), _lib.ptr(dq), dq.stride(0), _lib.ptr(dk),
          dk.stride(0), _lib.ptr(dv), dv.stride(0))


def tf_loss_fwd(logits, C, targets, pad_idx):
    """logits fp32 [B*T, ld>=C], targets int64 [B,T] -> (row_lse [B*T], acc [2] = (sum of NLL, counted rows))."""
    _chk(logits, F32, "logits")
    assert targets.dtype == I64 and targets.is_contiguous()
    B, T = targets.shape
    row_lse = torch.empty(B * T, dtype=F32, device=logits.device)
    acc = torch.empty(2, dtype=F32, device=logits.device)
    _call("ccd_tf_loss_fwd", _lib.ptr(logits), logits.stride(0), int(C), _lib.ptr(targets), B * T, T, int(pad_idx),
          _lib.ptr(row_lse), _lib.ptr(acc))
    return row_lse, acc


def tf_loss_bwd(logits, C, targets, pad_idx, row_lse, acc, upstream, ldd):
    """upstream: fp32 device scalar (the gradient arriving at the loss) or None for 1."""
    _chk(upstream, F32, "upstream")
    B, T = targets.shape
    d = torch.empty((B * T, ldd), dtype=BF16, device=logits.device)
    _call("ccd_tf_loss_bwd", _lib.ptr(logits), logits.stride(0), int(C), _lib.ptr(targets), B * T, T, int(pad_idx),
          _lib.ptr(row_lse), _lib.ptr(acc), _lib.ptr(upstream), _lib.ptr(d), ldd)
    return d


def greedy_step(logits, C, probs, step, seq):
    """probs fp32 [B, steps, C], seq int64 [B, seq_len]: writes probs[:, step] and seq[:, step+1]."""
    _chk(logits, F32, "logits"); _chk(probs, F32, "probs")
    assert seq.dtype == I64 and seq.is_contiguous() and probs.is_contiguous()
    B = seq.shape[0]
    _call("ccd_greedy_step", _lib.ptr(logits), logits.stride(0), int(C), B, _lib.ptr(probs), probs.shape[1], int(step),
          _lib.ptr(seq), seq.shape[1])
