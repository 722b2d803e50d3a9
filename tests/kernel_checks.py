"""Parity checks of individual HIP kernels against plain PyTorch fp32 math on the same (bf16-rounded) inputs.
Used by test_kernels_sim.py (CPU SIMT executor, tiny shapes) and test_kernels_gpu.py (-m gpu, real MI355X)."""
import math

import torch
import torch.nn.functional as F

from ccd_amd import ops

BF = torch.bfloat16


def rnd(shape, gen, scale=1.0):
    return torch.randn(shape, generator=gen) * scale


def close(got, want, rtol, atol, what):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.4g} "
                           f"(want max {want.abs().max().item():.4g})")


def check_gemm_nt(dev, M, N, K, seed=0):
    g = torch.Generator().manual_seed(seed)
    a = rnd((M, K), g).to(BF); b = rnd((N, K), g, 0.2).to(BF)
    bias = rnd((N,), g); resid = rnd((M, N), g)
    rows_per_sample = 16
    rowscale = (torch.rand((M + rows_per_sample - 1) // rows_per_sample, generator=g) > 0.3).float() * 1.25
    ref = a.float() @ b.float().t()
    A, B_, bias_d, resid_d, rs_d = a.to(dev), b.to(dev), bias.to(dev), resid.to(dev), rowscale.to(dev)
    # asymmetric operands + non-square shapes make a transposed C-write visible (guide rule 16)
    close(ops.gemm_nt(A, B_, epilogue=ops.EPI_BF16, bias=bias_d), ref + bias, 1e-2, 2e-2, "nt/bf16")
    close(ops.gemm_nt(A, B_, epilogue=ops.EPI_BF16), ref, 1e-2, 2e-2, "nt/bf16-nobias")
    close(ops.gemm_nt(A, B_, epilogue=ops.EPI_F32, bias=bias_d, alpha=0.5), 0.5 * ref + bias, 1e-4, 1e-4, "nt/f32")
    u, gl = ops.gemm_nt(A, B_, epilogue=ops.EPI_GELU, bias=bias_d)
    close(u, ref + bias, 1e-2, 2e-2, "nt/gelu-u")
    close(gl, F.gelu(ref + bias), 1e-2, 2e-2, "nt/gelu-g")
    want = resid + (ref + bias) * rowscale.repeat_interleave(rows_per_sample)[:M, None]
    close(ops.gemm_nt(A, B_, epilogue=ops.EPI_RESID, bias=bias_d, resid=resid_d, rowscale=rs_d,
                      rows_per_sample=rows_per_sample), want, 1e-4, 1e-4, "nt/resid")
    close(ops.gemm_nt(A, B_, epilogue=ops.EPI_RESID, bias=bias_d, resid=resid_d), resid + ref + bias, 1e-4, 1e-4,
          "nt/resid-noscale")
    aux = rnd((M, N), g).to(BF)
    xx = aux.float().double().requires_grad_(True)
    F.gelu(xx).sum().backward()
    close(ops.gemm_nt(A, B_, epilogue=ops.EPI_DGELU, aux=aux.to(dev)), ref * xx.grad.float(), 1e-2, 2e-2, "nt/dgelu")
    # strided A (a column slice of a wider buffer) and both tile orders
    wide = torch.zeros((M, K + 64), dtype=BF); wide[:, 64:] = a
    wd = wide.to(dev)
    close(ops.gemm_nt(wd[:, 64:], B_, epilogue=ops.EPI_F32, m_fastest=1), ref, 1e-4, 1e-4, "nt/strided-mfast")
    close(ops.gemm_nt(wd[:, 64:], B_, epilogue=ops.EPI_F32, m_fastest=0), ref, 1e-4, 1e-4, "nt/strided-nfast")


def check_gemm_tn(dev, Mc, P, Q, seed=1, splits=0):
    g = torch.Generator().manual_seed(seed)
    a = rnd((Mc, P), g).to(BF); b = rnd((Mc, Q), g).to(BF)
    ref = a.float().t() @ b.float()
    out = torch.full((P, Q), 0.5, dtype=torch.float32).to(dev)
    ops.gemm_tn(a.to(dev), b.to(dev), out, accumulate=True, splits=splits)
    tol = 1e-3 * math.sqrt(Mc)
    close(out, ref + 0.5, 1e-4, tol, "tn/atomic")
    out2 = torch.empty((P, Q), dtype=torch.float32).to(dev)
    ops.gemm_tn(a.to(dev), b.to(dev), out2, accumulate=False, alpha=2.0)
    close(out2, 2 * ref, 1e-4, 2 * tol, "tn/store")


def check_layernorm(dev, rows, E, seed=2):
    g = torch.Generator().manual_seed(seed)
    x = rnd((rows, E), g) * 2 + 0.3
    gamma = 1 + 0.1 * rnd((E,), g); beta = 0.1 * rnd((E,), g)
    y, mean, rstd = ops.ln_fwd(x.to(dev), gamma.to(dev), beta.to(dev), 1e-6)
    close(y, F.layer_norm(x, (E,), gamma, beta, 1e-6), 1e-2, 1e-2, "ln/y")
    close(mean, x.mean(1), 1e-5, 1e-5, "ln/mean")
    close(rstd, 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6), 1e-4, 1e-5, "ln/rstd")
    dy = rnd((rows, E), g).to(BF)
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    F.layer_norm(xr, (E,), gr, br, 1e-6).backward(dy.float())
    g0 = rnd((rows, E), g)
    for acc in (True, False):
        gbuf = g0.clone().to(dev)
        dgam = torch.zeros(E).to(dev); dbet = torch.zeros(E).to(dev)
        ops.ln_bwd(dy.to(dev), x.to(dev), mean, rstd, gamma.to(dev), gbuf, dgam, dbet, accumulate=acc)
        close(gbuf, xr.grad + (g0 if acc else 0), 1e-3, 1e-4, f"ln/dx acc={acc}")
        close(dgam, gr.grad, 1e-3, 1e-3, "ln/dgamma")
        close(dbet, br.grad, 1e-3, 1e-3, "ln/dbeta")


def attention_ref(qkv, heads):
    views, T, E3 = qkv.shape
    E = E3 // 3
    d = E // heads
    q, k, v = qkv.reshape(views, T, 3, heads, d).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q @ k.transpose(-2, -1)) * d ** -0.5, dim=-1)
    return (a @ v).transpose(1, 2).reshape(views, T, E)


def check_attention(dev, views, heads, seed=3, spike=False):
    g = torch.Generator().manual_seed(seed)
    E = heads * 64
    qkv = rnd((views, 256, 3 * E), g).to(BF)
    if spike:   # one dominant key per query row: exercises large score ranges
        qkv[:, :, :E] *= 4.0
    scale = 64 ** -0.5
    qf = qkv.float().requires_grad_(True)
    ref = attention_ref(qf, heads)
    out, lse = ops.attention_fwd(qkv.to(dev), heads, scale)
    close(out, ref, 2e-2, 2e-2, "attn/out")
    q, k, _ = qkv.float().reshape(views, 256, 3, heads, 64).permute(2, 0, 3, 1, 4)
    close(lse, torch.logsumexp((q @ k.transpose(-2, -1)) * scale, dim=-1), 1e-3, 2e-3, "attn/lse")
    d_out = rnd((views, 256, E), g).to(BF)
    ref.backward(d_out.float())
    d_qkv = ops.attention_bwd(qkv.to(dev), out, d_out.to(dev), lse, heads, scale)
    want = qf.grad
    for i, nm in enumerate("qkv"):
        close(d_qkv[..., i * E:(i + 1) * E], want[..., i * E:(i + 1) * E], 4e-2, 4e-2 * want.abs().max().item(),
              f"attn/d{nm}")
