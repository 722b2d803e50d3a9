"""Thin tensor-level wrappers over the C ABI (include/ccd_hip.h).  Every function enqueues HIP kernels on the
current stream; nothing here computes on the host."""
from __future__ import annotations

import torch

from . import _lib

EPI_BF16, EPI_GELU, EPI_RESID, EPI_F32, EPI_ATOMIC, EPI_DGELU = range(6)
BF16, F32 = torch.bfloat16, torch.float32


def _chk(t, dtype, name):
    if t is None:
        return
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def gemm_nt(a, b, *, epilogue=EPI_BF16, out=None, out2=None, bias=None, resid=None, rowscale=None,
            rows_per_sample=1, aux=None, alpha=1.0, m_fastest=None):
    """out[M,N] = a[M,K] @ b[N,K]^T with a fused epilogue (see include/ccd_hip.h)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(bias, F32, "bias"); _chk(resid, F32, "resid")
    _chk(rowscale, F32, "rowscale"); _chk(aux, BF16, "aux")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    odt = F32 if epilogue in (EPI_RESID, EPI_F32, EPI_ATOMIC) else BF16
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=a.device)
    _chk(out, odt, "out")
    if epilogue == EPI_GELU and out2 is None:
        out2 = torch.empty((M, N), dtype=BF16, device=a.device)
    if m_fastest is None:
        m_fastest = 1 if N > M else 0
    lib = _lib.get()
    _lib.check(lib.ccd_gemm_nt(_lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), M, N, K, epilogue, _lib.ptr(out),
                               out.stride(0), _lib.ptr(out2), 0 if out2 is None else out2.stride(0), _lib.ptr(bias),
                               _lib.ptr(resid), 0 if resid is None else resid.stride(0), _lib.ptr(rowscale),
                               rows_per_sample, _lib.ptr(aux), 0 if aux is None else aux.stride(0), float(alpha),
                               int(m_fastest), _lib.stream()), "gemm_nt")
    return (out, out2) if epilogue == EPI_GELU else out


def gemm_tn(a, b, out, *, accumulate=True, alpha=1.0, splits=0):
    """out[P,Q] (+)= a[Mc,P]^T @ b[Mc,Q]  (fp32 out; accumulate=True adds with fp32 atomics, split over Mc)."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(out, F32, "out")
    Mc, Pd = a.shape
    Q = b.shape[1]
    assert b.shape[0] == Mc and tuple(out.shape) == (Pd, Q)
    lib = _lib.get()
    _lib.check(lib.ccd_gemm_tn(_lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), Pd, Q, Mc,
                               EPI_ATOMIC if accumulate else EPI_F32, _lib.ptr(out), out.stride(0), float(alpha),
                               int(splits) if accumulate else 1, _lib.stream()), "gemm_tn")
    return out


def ln_fwd(x, gamma, beta, eps=1e-6):
    """x [rows,E] fp32 -> (y bf16, mean, rstd)."""
    _chk(x, F32, "x")
    rows, E = x.shape
    y = torch.empty((rows, E), dtype=BF16, device=x.device)
    mean = torch.empty(rows, dtype=F32, device=x.device)
    rstd = torch.empty(rows, dtype=F32, device=x.device)
    _lib.check(_lib.get().ccd_ln_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), _lib.ptr(mean),
                                     _lib.ptr(rstd), rows, E, float(eps), _lib.stream()), "ln_fwd")
    return y, mean, rstd


def ln_bwd(dy, x, mean, rstd, gamma, g, dgamma, dbeta, accumulate=True):
    """g (+)= LN'(dy); dgamma += , dbeta += (in place)."""
    _chk(dy, BF16, "dy"); _chk(x, F32, "x"); _chk(g, F32, "g")
    rows, E = x.shape
    _lib.check(_lib.get().ccd_ln_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma),
                                     _lib.ptr(g), 1 if accumulate else 0, _lib.ptr(dgamma), _lib.ptr(dbeta), rows, E,
                                     _lib.stream()), "ln_bwd")
    return g


def attention_fwd(qkv, heads, scale):
    """qkv [views,256,3*E] bf16 -> (out [views,256,E] bf16, lse [views,heads,256] fp32)."""
    _chk(qkv, BF16, "qkv")
    views, T, E3 = qkv.shape
    assert T == 256 and E3 == 3 * heads * 64 and qkv.is_contiguous()
    out = torch.empty((views, T, E3 // 3), dtype=BF16, device=qkv.device)
    lse = torch.empty((views, heads, T), dtype=F32, device=qkv.device)
    _lib.check(_lib.get().ccd_attention_fwd(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse), views, heads, float(scale),
                                            _lib.stream()), "attention_fwd")
    return out, lse


def attention_bwd(qkv, out, d_out, lse, heads, scale):
    _chk(qkv, BF16, "qkv"); _chk(out, BF16, "out"); _chk(d_out, BF16, "d_out")
    assert qkv.is_contiguous() and out.is_contiguous() and d_out.is_contiguous()
    views = qkv.shape[0]
    d_qkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    _lib.check(_lib.get().ccd_attention_bwd(_lib.ptr(qkv), _lib.ptr(out), _lib.ptr(d_out), _lib.ptr(lse),
                                            _lib.ptr(delta), _lib.ptr(d_qkv), views, heads, float(scale),
                                            _lib.stream()), "attention_bwd")
    return d_qkv
