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
    bad = ~(err <= tol)                     # (NaNs are bad: the executor poisons registers / LDS that are read too early)
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
    g2 = torch.empty((M, N), dtype=BF).to(dev)         # optional second output: gelu(aux), for the weight-gradient product
    close(ops.gemm_nt(A, B_, epilogue=ops.EPI_DGELU, aux=aux.to(dev), out2=g2), ref * xx.grad.float(), 1e-2, 2e-2, "nt/dgelu+g")
    close(g2, F.gelu(aux.float()), 1e-2, 1e-3, "nt/dgelu-gelu(u)")
    cs = torch.full((N,), 3.0).to(dev)
    du = ops.gemm_nt(A, B_, epilogue=ops.EPI_DGELU, aux=aux.to(dev), colsum=cs)
    rms_d = float((ref * xx.grad.float()).pow(2).mean().sqrt())
    close(cs, (ref * xx.grad.float()).sum(0) + 3.0, 1e-3, max(2e-2, 1e-2 * rms_d) * (M ** 0.5), "nt/dgelu-colsum")
    cs = torch.zeros(N).to(dev)
    ops.gemm_nt(A, B_, epilogue=ops.EPI_BF16, bias=bias_d, colsum=cs)
    # column sums may be taken over the bf16-rounded outputs: noise ~ 2^-9 * rms * sqrt(M) per column
    rms = float((ref + bias).pow(2).mean().sqrt())
    close(cs, (ref + bias).sum(0), 1e-3, max(2e-2, 1e-2 * rms) * (M ** 0.5), "nt/bf16-colsum")
    none_u, g_only = ops.gemm_nt(A, B_, epilogue=ops.EPI_GELU, bias=bias_d, store_u=False)
    assert none_u is None
    close(g_only, F.gelu(ref + bias), 1e-2, 2e-2, "nt/gelu-only")
    # strided A (a column slice of a wider buffer) and both tile orders
    wide = torch.zeros((M, K + 64), dtype=BF); wide[:, 64:] = a
    wd = wide.to(dev)
    close(ops.gemm_nt(wd[:, 64:], B_, epilogue=ops.EPI_F32, m_fastest=1), ref, 1e-4, 1e-4, "nt/strided-mfast")
    close(ops.gemm_nt(wd[:, 64:], B_, epilogue=ops.EPI_F32, m_fastest=0), ref, 1e-4, 1e-4, "nt/strided-nfast")


def check_gemm_dynamic_rows(dev, M, N, K, live, seed=5):
    """Device-side row count: only rows < 2*live are computed, the rest of the output is left untouched."""
    g = torch.Generator().manual_seed(seed)
    a = rnd((M, K), g).to(BF).to(dev); b = rnd((N, K), g, 0.2).to(BF).to(dev)
    d_rows = torch.tensor([live], dtype=torch.int32, device=dev)
    out = torch.full((M, N), 7.0, device=dev)
    ops.gemm_nt(a, b, epilogue=ops.EPI_F32, out=out, d_rows=d_rows, rows_mul=2)
    close(out[:2 * live], a[:2 * live].float() @ b.float().t(), 1e-4, 1e-3, "dyn/nt")
    assert bool((out[2 * live:] == 7.0).all()), "rows past the device-side count were written"
    outb = torch.full((M, N), 7.0, device=dev).to(BF)
    ops.gemm_nt(a, b, epilogue=ops.EPI_BF16, out=outb, d_rows=d_rows, rows_mul=2)
    close(outb[:2 * live], a[:2 * live].float() @ b.float().t(), 1e-2, 2e-2, "dyn/nt-bf16")
    assert bool((outb[2 * live:].float() == 7.0).all()), "rows past the device-side count were written (bf16)"


def check_gemm_tn_colsum(dev, Mc=300, P=264, Q=72, splits=3, seed=12):
    """Weight + bias gradient in one pass: C += A^T B and colsum += A.sum(0), several column tiles (only the first carries the
    sums), ragged contraction slices, accumulation onto existing values."""
    g = torch.Generator().manual_seed(seed)
    a = rnd((Mc, P), g).to(BF); b = rnd((Mc, Q), g).to(BF)
    c0 = rnd((P, Q), g); s0 = rnd((P,), g)
    out, cs = c0.clone().to(dev), s0.clone().to(dev)
    ops.gemm_tn_colsum(a.to(dev), b.to(dev), out, cs, splits=splits)
    close(out, c0 + a.float().t() @ b.float(), 2e-3, 2e-3 * Mc ** 0.5, "gemm_tn_colsum/C")
    close(cs, s0 + a.float().sum(0), 1e-4, 1e-4 * Mc ** 0.5, "gemm_tn_colsum/colsum")


def check_rowproj(dev, M, N, K, seed=0, strided=True):
    """rowproj.h through ccd_gemm_nt (EPI_BF16, K in {384, 512}, N % 64 == 0, policy rowproj): activation rows resident in
    registers, weights through the LDS ring.  Ragged last tile, strided A / out (the qkv buffer's column slices), with and
    without bias; the result must equal the tiled kernels' to bf16 rounding and rows beyond M must stay untouched."""
    g = torch.Generator().manual_seed(seed)
    a_full = rnd((M, K + 64), g).to(BF)
    a = a_full[:, 8:8 + K] if strided else a_full[:, :K].contiguous()
    b = rnd((N, K), g, 0.2).to(BF)
    bias = rnd((N,), g)
    ref = a.float() @ b.float().t()
    A = a_full.to(dev)[:, 8:8 + K] if strided else a.to(dev)
    B_ = b.to(dev)
    with ops.policy(rowproj=1, rowproj_min_m=1):
        got = ops.gemm_nt(A, B_, epilogue=ops.EPI_BF16, bias=bias.to(dev))
        close(got, ref + bias, 1e-2, 2e-2, "rowproj/bias")
        big = torch.full((M + 3, N + 64), 7.0, dtype=BF).to(dev)
        out = big[:M, 32:32 + N]
        ops.gemm_nt(A, B_, epilogue=ops.EPI_BF16, out=out)
        close(out, ref, 1e-2, 2e-2, "rowproj/nobias-strided-out")
        untouched = big.clone()
        untouched[:M, 32:32 + N] = 7.0
        assert bool((untouched == 7.0).all()), "rowproj wrote outside its [M, N] block"
    with ops.policy(rowproj=0):
        tiled = ops.gemm_nt(A, B_, epilogue=ops.EPI_BF16, bias=bias.to(dev))
    close(got, tiled.float(), 1e-2, 1e-2, "rowproj vs tiled kernel")


def check_gemm_nt_split_k(dev, M=200, N=72, K=16384 + 8192 + 64, seed=9):
    """EPI_ATOMIC on the NT product: C += A . B^T with the contraction cut into slices of 8192 (the head's data gradient,
    K = 65536), ragged last slice, accumulation onto existing values."""
    g = torch.Generator().manual_seed(seed)
    a = rnd((M, K), g, 0.05).to(BF); b = rnd((N, K), g, 0.05).to(BF)
    base = rnd((M, N), g)
    out = base.clone().to(dev)
    ops.gemm_nt(a.to(dev), b.to(dev), epilogue=ops.EPI_ATOMIC, out=out)
    want = base + a.float() @ b.float().t()
    close(out, want, 2e-3, 2e-3 * float(want.abs().max()), "gemm_nt split-K atomic")


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


def check_gemm_tn_pair(dev, Mc, shape1, shape2, seed=5):
    """Two weight-gradient products over the same rows in one call (ccd_gemm_tn_pair): accumulation onto existing values."""
    g = torch.Generator().manual_seed(seed)
    bases, wants, args = [], [], []
    for P, Q in (shape1, shape2):
        a = rnd((Mc, P), g).to(BF); b = rnd((Mc, Q), g).to(BF)
        base = rnd((P, Q), g)
        bases.append(base); wants.append(base + a.float().t() @ b.float()); args.append((a.to(dev), b.to(dev)))
    tol = 1e-3 * math.sqrt(Mc)
    # split-K workspace (per-slice partial tiles + reduction pass; round 4) and the fp32-atomic epilogue
    for workspace in (True, False):
        outs = [b.clone().to(dev) for b in bases]
        ops.gemm_tn_pair(args[0][0], args[0][1], outs[0], args[1][0], args[1][1], outs[1], workspace=workspace)
        close(outs[0], wants[0], 1e-4, tol, f"tn_pair/first ws={workspace}")
        close(outs[1], wants[1], 1e-4, tol, f"tn_pair/second ws={workspace}")
        # (on the GPU, gemm_tn384.h's shapes; a pair that does not fit the chip's slots - the CPU executor's 2-CU chips - falls back to
        # single products that add by atomics)
        if workspace and dev.type == "cuda" and ops.policy_get("gemm_tn384") and all(P % 384 == 0 and Q % 192 == 0 for P, Q in (shape1, shape2)):
            # plain stores + one reduction in a fixed order: the same bits on every run
            again = [b.clone().to(dev) for b in bases]
            ops.gemm_tn_pair(args[0][0], args[0][1], again[0], args[1][0], args[1][1], again[1], workspace=True)
            assert torch.equal(again[0], outs[0]) and torch.equal(again[1], outs[1]), "tn_pair with a workspace is not deterministic"


def check_layernorm(dev, rows, E, seed=2, g16=False):
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
    if g16:                      # the gradient stream as a bf16 tensor (ccd_ln_bwd_g16)
        g0 = g0.to(BF).float()
    rows_per_sample = 4
    rowscale = (torch.rand((rows + rows_per_sample - 1) // rows_per_sample, generator=g) > 0.3).float() * 1.25
    for acc in (True, False):
        for fused in (False, True):
            gbuf = (g0.to(BF) if g16 else g0.clone()).to(dev)
            dgam = torch.zeros(E).to(dev); dbet = torch.zeros(E).to(dev)
            gb = torch.zeros(rows, E, dtype=BF).to(dev); dbias = torch.full((E,), 0.25).to(dev)
            kw = dict(gb=gb, rowscale=rowscale.to(dev), rows_per_sample=rows_per_sample, dbias=dbias) if fused else {}
            ops.ln_bwd(dy.to(dev), x.to(dev), mean, rstd, gamma.to(dev), gbuf, dgam, dbet, accumulate=acc, **kw)
            want_g = xr.grad + (g0 if acc else 0)
            close(gbuf, want_g, 1e-2 if g16 else 1e-3, 1e-3 if g16 else 1e-4, f"ln/dx acc={acc}")
            close(dgam, gr.grad, 1e-3, 1e-3, "ln/dgamma")
            close(dbet, br.grad, 1e-3, 1e-3, "ln/dbeta")
            if fused:
                want_gb = (want_g * rowscale.repeat_interleave(rows_per_sample)[:rows, None]).to(BF)
                close(gb, want_gb, 1e-2, 1e-3, "ln/gb")
                close(dbias, gb.float().sum(0).cpu() + 0.25, 1e-4, 1e-3, "ln/dbias")


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
    # the qkv-bias gradient without a pass over d_qkv: q part from the dQ kernel's fp32 tiles, k part identically zero (nothing
    # added), v part = colsum(d_out) handed in - once as the vector itself, once factored as vec @ mat (d_out = gb @ mat)
    d_bias = torch.full((3 * E,), 0.5, dtype=torch.float32, device=dev)          # accumulated into, not overwritten
    dcs = d_out.float().sum((0, 1)).to(dev)
    d_qkv = ops.attention_bwd(qkv.to(dev), out, d_out.to(dev), lse, heads, scale, d_bias=d_bias, dout_colsum=dcs)
    want = qf.grad
    want_b = want.reshape(-1, 3 * E).double().sum(0).float()
    # (rounding errors of the bf16 operands add up like a random walk over the rows: measured 0.02 at 768 rows, |element| < 1)
    tol_b = 4e-3 * (views * 256) ** 0.5 * want.abs().max().item()
    got_b = d_bias.cpu() - 0.5
    close(got_b[:E], want_b[:E], 1e-2, tol_b, "attn/d_qkv_bias q")
    close(got_b[2 * E:], want_b[2 * E:], 1e-2, tol_b, "attn/d_qkv_bias v")
    assert bool((got_b[E:2 * E] == 0).all()) and want_b[E:2 * E].abs().max().item() < tol_b, "the key bias has no gradient"
    vec = rnd((E,), g).to(dev)
    mat = (torch.outer(vec.cpu(), dcs.cpu()) / vec.cpu().pow(2).sum() + 0.0).to(dev)      # vec @ mat == dcs
    d_bias2 = torch.zeros(3 * E, dtype=torch.float32, device=dev)
    ops.attention_bwd(qkv.to(dev), out, d_out.to(dev), lse, heads, scale, d_bias=d_bias2, dout_colsum=vec, dout_colsum_mat=mat)
    close(d_bias2.cpu()[2 * E:], dcs.cpu(), 1e-4, 1e-4 * dcs.abs().max().item(), "attn/d_qkv_bias v (factored)")
    close(d_bias2.cpu()[:E], got_b[:E], 1e-5, 1e-5 * tol_b, "attn/d_qkv_bias q (second launch)")
    d_plain = ops.attention_bwd(qkv.to(dev), out, d_out.to(dev), lse, heads, scale)      # without the bias gradient: same d_qkv
    assert torch.equal(d_plain, d_qkv)
    for i, nm in enumerate("qkv"):
        close(d_qkv[..., i * E:(i + 1) * E], want[..., i * E:(i + 1) * E], 4e-2, 4e-2 * want.abs().max().item(),
              f"attn/d{nm}")


# =========================================================================================== round-2 kernels
import os
import struct

import numpy as np

from oracle import ccd_oracle as O
from oracle import ccl_np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_ccl(dev):
    g = np.load(os.path.join(GOLD, "ccl_cases.npz"))
    masks = torch.from_numpy(g["masks"].astype(np.float32)).to(dev)
    got = ops.ccl_label(masks).cpu().numpy()
    for name, out, want, tie in zip(g["names"], got, g["idmaps"], g["has_tie"]):
        if not tie:
            np.testing.assert_array_equal(out, want, err_msg=str(name))
        else:
            np.testing.assert_array_equal(out, ccl_np.label_idmap(g["masks"][list(g["names"]).index(name)]))
    rs = np.random.RandomState(5)
    rnd_masks = (rs.uniform(size=(24, 32, 128)) < rs.uniform(0.2, 0.75, size=(24, 1, 1))).astype(np.float32)
    got = ops.ccl_label(torch.from_numpy(rnd_masks).to(dev)).cpu().numpy()
    for m, o in zip(rnd_masks, got):
        np.testing.assert_array_equal(o, ccl_np.label_idmap(m))


def check_warp(dev):
    g = np.load(os.path.join(GOLD, "small_step.npz"))
    for p in ("s0/", "s1/"):
        ids = g[p + "zero_idmap"]
        B = ids.shape[0] // 2
        theta = torch.from_numpy(g[p + "metrics"]).to(dev)
        got = ops.warp_idmap(torch.from_numpy(ids[:B].copy()).to(dev), theta).cpu().numpy()
        np.testing.assert_array_equal(got, ids[B:], err_msg=p + "clusters")
        m = torch.from_numpy(g[p + "masks"].astype(np.float32)).to(dev)
        mi = ops.warp_idmap(ops.mask_to_idmap(m), theta).cpu().numpy()
        np.testing.assert_array_equal((mi != 255).astype(np.uint8), g[p + "masks_image"], err_msg=p + "gt")
    ids = g["pred/zero_idmap"]
    B = ids.shape[0] // 2
    mask = torch.from_numpy(g["pred/mask"].astype(np.float32)).to(dev)
    src = ops.ccl_label(mask)
    np.testing.assert_array_equal(src.cpu().numpy(), ids[:B])
    got = ops.warp_idmap(src, torch.from_numpy(g["pred/metrics"]).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(got, ids[B:])


def _golden_kmeans():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kmeans_masks.npz"))
    grays, masks, off = [], [], 0
    for h, w in z["hw"]:
        grays.append(z["gray"][off:off + h * w].reshape(h, w)); masks.append(z["mask"][off:off + h * w].reshape(h, w))
        off += h * w
    return grays, masks


def check_kmeans2_mask(dev, seed=51, extra=12):
    """clusterpixels(im, 2) on the device (mask_create/generate_mask.py:13-29): bit-equal to the REAL function's outputs on
    the committed word images, and to the oracle on ragged random batches incl. the degenerate cases."""
    from oracle import datapipe_np as D
    grays, masks = _golden_kmeans()
    got = ops.kmeans2_mask(grays, device=dev)
    for i, (g, m) in enumerate(zip(got, masks)):
        np.testing.assert_array_equal(g, m, err_msg=f"fixture image {i}")
    rs = np.random.RandomState(seed)
    batch = [np.full((5, 7), 93, np.uint8),                                   # one gray level: no clustering, all zero
             np.array([[0, 255]], np.uint8),                                  # 1 x 2
             (np.arange(6 * 300) % 251).astype(np.uint8).reshape(6, 300),      # flat histogram
             np.pad(np.full((10, 20), 200, np.uint8), 6, constant_values=30),  # bright blob on dark: no flip
             np.pad(np.full((10, 20), 30, np.uint8), 6, constant_values=200)]  # dark blob on bright: flip
    for _ in range(extra):
        h, w = rs.randint(1, 70), rs.randint(1, 300)
        a = rs.normal(rs.choice([60, 180]), 25, size=(h, w))
        a[rs.uniform(size=(h, w)) < 0.3] += rs.choice([-90, 90])
        batch.append(np.clip(np.rint(a), 0, 255).astype(np.uint8))
    got = ops.kmeans2_mask(batch, device=dev)
    for i, (g, a) in enumerate(zip(got, batch)):
        np.testing.assert_array_equal(g, D.kmeans2_mask(a), err_msg=f"random image {i} {a.shape}")
    assert got[3][8, 10] == 1 and got[3][0, 0] == 0 and got[4][8, 10] == 1 and got[4][0, 0] == 0
    assert ops.kmeans2_mask([], device=dev) == []


def _aug_params(seed=1, preinv=0, a=0, a_args=(), a_kern=None, b=0, b_args=(), c=0, c_args=(), blur_kern=None, d=0, d_args=()):
    """One parameter row (kernels/datapipe.h layout): member + arguments of each group; `a_kern` = the 3 x 3 kernel of an
    `arithmetic` filter member, `blur_kern` = a correlation kernel embedded in the `Blur` group's 7 x 7 grid."""
    from ccd_amd.dataset import augment as A
    p = A.IDENTITY_PARAMS.copy()
    p[A.P_SEED], p[A.P_PREINV] = seed, preinv
    p[A.P_A], p[A.P_B], p[A.P_C], p[A.P_D] = a, b, c, d
    p[A.P_A + 1:A.P_A + 1 + len(a_args)] = a_args
    p[A.P_B + 1:A.P_B + 1 + len(b_args)] = b_args
    p[A.P_C + 1:A.P_C + 1 + len(c_args)] = c_args
    p[A.P_D + 1:A.P_D + 1 + len(d_args)] = d_args
    if a_kern is not None:
        p[A.P_AK:A.P_AK + 9] = np.asarray(a_kern, np.float32).reshape(-1)
    if blur_kern is not None:
        A._set_filter(p, blur_kern)
    return p


def _aug_member_rows():
    """Every member of the chain at least once (alone), then combinations across the groups."""
    from ccd_amd.dataset import augment as A
    R = _aug_params
    emboss = A._blend(0.6, [[-2.1, -1.1, 0], [-1.1, 1, 1.1], [0, 1.1, 2.1]])
    rows = [
        R(),                                                               # 0: identity (sample 0, both views)
        R(),
        R(seed=11, a=A.A_ADD_ELEM, a_args=(40, 0)), R(seed=12, a=A.A_ADD_ELEM, a_args=(40, 1)),
        R(seed=13, a=A.A_GAUSS, a_args=(30.0, 0)), R(seed=14, a=A.A_GAUSS, a_args=(8.0, 1)),
        R(seed=15, a=A.A_LAPLACE, a_args=(25.0, 0)), R(seed=16, a=A.A_LAPLACE, a_args=(5.0, 1)),
        R(seed=17, a=A.A_POISSON, a_args=(31.0, 0)), R(seed=18, a=A.A_POISSON, a_args=(2.5, 1)),
        R(seed=19, a=A.A_MUL, a_args=(1.3, 0.7, 1.1)), R(seed=20, a=A.A_MUL_ELEM, a_args=(0.5, 1, 1.5)),
        R(seed=21, a=A.A_MUL_ELEM, a_args=(0.5, 0, 1.5)), R(seed=22, a=A.A_DROPOUT, a_args=(0.08, 0)),
        R(seed=23, a=A.A_DROPOUT, a_args=(0.05, 1)), R(seed=24, a=A.A_COARSE, a_args=(0.2, 0, 5, 19)),
        R(seed=25, a=A.A_COARSE, a_args=(0.2, 1, 3, 6)), R(seed=26, a=A.A_DROP2D, a_args=(5,)),
        R(seed=27, a=A.A_REPLACE, a_args=(0.1, 1, 0)), R(seed=28, a=A.A_REPLACE, a_args=(0.1, 0, 0)),
        R(seed=29, a=A.A_REPLACE, a_args=(0.1, 0, 1)), R(seed=30, a=A.A_REPLACE, a_args=(0.1, 0, 2)),
        R(seed=31, a=A.A_INVERT), R(seed=32, a=A.A_SOLARIZE, a_args=(100.0,)),
        R(seed=33, a=A.A_JPEG, a_args=(2,)), R(seed=34, a=A.A_JPEG, a_args=(31,)),
        R(seed=35, a=A.A_FILTER, a_kern=emboss), R(seed=36, a=A.A_FILTER, a_kern=A.directed_edge_kernel(0.8, 0.3)),
        R(seed=37, a=A.A_PILFILTER, a_args=(0.0, 1.0), a_kern=[-1, -1, -1, -1, 9, -1, -1, -1, -1]),
        R(seed=38, a=A.A_PILFILTER, a_args=(255.0, 1.0), a_kern=[-1, -1, -1, -1, 8, -1, -1, -1, -1]),
        R(seed=39, b=A.B_HUE_ADD, b_args=(37,)), R(seed=40, b=A.B_HUE_ADD, b_args=(100,)),
        R(seed=41, b=A.B_BRIGHT, b_args=(1.4, -20.0)), R(seed=42, b=A.B_MUL_HS, b_args=(1.37, 0.6)),
        R(seed=43, b=A.B_MUL_HS, b_args=(0.55, 1.45)), R(seed=44, b=A.B_ADD_HS, b_args=(-35, 40)),
        R(seed=45, b=A.B_ADD_HS, b_args=(21, -50)), R(seed=46, b=A.B_GRAY, b_args=(0.65,)),
        R(seed=77, b=A.B_KMEANS, b_args=(2,)), R(seed=78, b=A.B_KMEANS, b_args=(5,)), R(seed=79, b=A.B_KMEANS, b_args=(16,)),
        R(seed=80, b=A.B_KMEANS, b_args=(9,)),       # (an even number of new rows: the chains below keep their place on view 1 / view 2)
        R(seed=47, b=A.B_UNIFORM_Q, b_args=(2,)), R(seed=48, b=A.B_UNIFORM_Q, b_args=(11,)),
        R(seed=49, b=A.B_GAINS, b_args=(1.2, 1.0, 0.8)), R(seed=50, b=A.B_SHUFFLE, b_args=(4,)),
        R(seed=51, blur_kern=A.gaussian_kernel5(0.8)), R(seed=52, blur_kern=np.full((6, 6), 1 / 36.0)),      # AverageBlur k = 6: offsets -3 .. 2
        R(seed=53, blur_kern=np.full((2, 2), 0.25)), R(seed=54, blur_kern=A.motion_kernel(5, 37.0, -0.4)),
        R(seed=55, blur_kern=A._blend(0.4, [[-1, -1, -1], [-1, 8.3, -1], [-1, -1, -1]])), R(seed=56, c=A.C_MEDIAN, c_args=(3,)),
        R(seed=57, c=A.C_MEDIAN, c_args=(5,)), R(seed=58, c=A.C_MEDIAN, c_args=(7,)),
        R(seed=59, c=A.C_BILATERAL, c_args=(9, 60.0, 120.0)), R(seed=60, c=A.C_BILATERAL, c_args=(4, 15.0, 10.0)),
        R(seed=61, d=A.D_GAMMA, d_args=(1.7,)), R(seed=62, d=A.D_GAMMA, d_args=(0.55,)),
        R(seed=63, d=A.D_LINEAR, d_args=(0.6,)), R(seed=64, d=A.D_SIGMOID, d_args=(7.0, 0.45)),
        R(seed=65, d=A.D_LOG, d_args=(1.3,)), R(seed=66, d=A.D_HISTEQ_ALL),
        R(seed=71, d=A.D_HISTEQ_LAB), R(seed=72, d=A.D_CLAHE_ALL, d_args=(2.5, 4)),        # 4 tiles a side: both sides divide
        R(seed=73, d=A.D_CLAHE_ALL, d_args=(0.1, 12)), R(seed=74, d=A.D_CLAHE_ALL, d_args=(8.0, 3)),   # 12 / 3 tiles: padded (32 = 2 * 12 + 8)
        R(seed=75, d=A.D_CLAHE_LAB, d_args=(4.0, 7)), R(seed=76, d=A.D_CLAHE_LAB, d_args=(1.0, 5)),
        # the chain: arithmetic -> color -> Blur -> contrast on one image, with the leading Invert of the finetuning pipeline
        R(seed=67, preinv=1, a=A.A_GAUSS, a_args=(12.0, 0), b=A.B_MUL_HS, b_args=(1.2, 0.8), blur_kern=A.gaussian_kernel5(1.2),
          d=A.D_SIGMOID, d_args=(5.0, 0.5)),
        R(seed=68, a=A.A_JPEG, a_args=(17,), b=A.B_GRAY, b_args=(0.3,), c=A.C_MEDIAN, c_args=(3,), d=A.D_HISTEQ_ALL),
        R(seed=69, a=A.A_FILTER, a_kern=emboss, b=A.B_ADD_HS, b_args=(10, 10), c=A.C_BILATERAL, c_args=(5, 40.0, 40.0),
          d=A.D_LINEAR, d_args=(0.8,)),
        R(seed=70, a=A.A_DROPOUT, a_args=(0.05, 0), b=A.B_UNIFORM_Q, b_args=(6,), blur_kern=np.full((3, 3), 1 / 9.0),
          d=A.D_GAMMA, d_args=(1.3,)),
    ]
    return np.stack(rows).astype(np.float32)


def check_augment_views(dev, H=32, W=128, seed=52, max_samples=None):
    """The three views of a sample (datasetsupervised_kmeans.py:48-87) vs the numpy restatement (oracle/datapipe_np.py), EVERY
    member of the reference's chain exercised alone and in combination (same counter-based random numbers on both sides);
    geometry: view 2 of an un-augmented sample == F.grid_sample of view 0's pixels in the dataset's (size-1)-normalised
    convention, and identity theta reproduces the augmented view on view 2."""
    from oracle import datapipe_np as D
    from ccd_amd.dataset import augment as A
    rs = np.random.RandomState(seed)
    rows = _aug_member_rows()
    if max_samples is not None:                                        # the CPU executor's run: the first rows + the chains
        rows = np.concatenate([rows[:2 * max_samples - 4], rows[-4:]])
    B = (len(rows) + 1) // 2
    params = np.tile(A.IDENTITY_PARAMS, (B, 2, 1)).astype(np.float32)
    params.reshape(-1, A.AUG_NP)[:len(rows)] = rows
    # `weather`: cloud layers drawn by the product's host side (ccd_amd/dataset/weather.py) - Fog alone on a plain row, Clouds (one
    # or two layers) behind a whole chain, Fog on a view-2 row; the blend is what is compared, on the same fp16 planes
    from ccd_amd.dataset import weather as Wt
    ov = Wt.Overlays(H, W)
    flat = params.reshape(-1, A.AUG_NP)
    wrows = {}
    for r, name in ((6, "Fog"), (len(rows) - 2, "Clouds"), (9, "Fog"), (12, "Snowflakes"), (15, "Rain"), (len(rows) - 4, "Snowflakes")):
        if r >= len(flat):
            continue
        flat[r, A.P_W], flat[r, A.P_W + 1] = -1, ov.add_task(name, int(rs.randint(0, 1 << 31)))
        flat[r, A.P_W + 2] = Wt.SNOW_MODE if name == "Snowflakes" else Wt.CLOUD_MODE
        wrows[r] = name
    planes = ov.resolve(params, A.P_W)
    assert planes is not None and planes.dtype == np.float16 and (flat[list(wrows), A.P_W] >= 1).all()
    img = rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    for b in range(B):                                                 # text-like structure under the noise: blocks of colour
        img[b] = (0.35 * img[b] + 0.65 * np.array(rs.randint(0, 256, 3))).astype(np.uint8)
        for _ in range(5):
            y0, x0 = rs.randint(0, H - 6), rs.randint(0, W - 10)
            img[b, y0:y0 + rs.randint(3, H // 2), x0:x0 + rs.randint(3, 12)] = rs.randint(0, 256, 3)
    img[1, :, :W // 2] = img[1, 0, 0]                                  # a flat half (the HSV / histogram members' degenerate inputs)
    theta = A.sample_theta(rs, B, H, W, p_warp=1.0)
    theta[2] = np.eye(3)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    # piecewise-affine warps (the finetuning geometry): samples 3 and 5 take a host-drawn source-position map instead of theta
    wm = Wt.WarpMaps(H, W)
    warp_rows = [b for b in (3, 5) if b < B]
    for b in warp_rows:
        params[b, 1, A.P_WARP] = wm.park("PiecewiseAffine", int(rs.randint(0, 1 << 31)))
    maps = wm.resolve(params, A.P_WARP)
    assert maps is not None and maps.dtype == np.float32 and maps.shape == (len(warp_rows), 2, H, W)
    assert [int(params[b, 1, A.P_WARP]) for b in warp_rows] == list(range(1, len(warp_rows) + 1))
    got = ops.augment_views(torch.from_numpy(img).to(dev), torch.from_numpy(params).to(dev), torch.from_numpy(theta).to(dev),
                            mean, std, overlay=torch.from_numpy(planes).to(dev), warp_maps=torch.from_numpy(maps).to(dev)).cpu().numpy()
    assert got.shape == (B, 3, 3, H, W) and np.isfinite(got).all()
    plain = ops.augment_views(torch.from_numpy(img).to(dev), torch.from_numpy(params).to(dev), torch.from_numpy(theta).to(dev),
                              mean, std).cpu().numpy()           # without the planes the rows' weather entries are ignored (and the maps: theta)
    for b in warp_rows:
        assert np.abs(got[b, 2] - plain[b, 2]).max() > 0.05, (b, "the warp map changed nothing")
    for r in wrows:
        assert np.abs(got[r // 2, 1 + r % 2] - plain[r // 2, 1 + r % 2]).max() > 0.05, (r, "the cloud layers changed nothing")
    # view 0 is the plain normalised image, exactly
    v0 = ((img.astype(np.float32) * np.float32(1 / 255.0) - np.float32(mean)) * (np.float32(1) / np.float32(std))).transpose(0, 3, 1, 2)
    np.testing.assert_allclose(got[:, 0], v0, rtol=0, atol=1e-6)
    # view 1 of every sample = the restated chain of its row, to the LEVEL.  powf / logf / expf / sinf differ by ulps between libm
    # and the device, and a value on an x.5 tie (or a noise draw on a decision boundary, a DCT coefficient on a quantisation
    # step) rounds either way: a small share of pixels may be one level off, JPEG blocks a few levels
    istd255 = 255.0 * np.float32(std)[:, None, None]
    for b in range(B):
        p = params[b, 0]
        stg = D.staged_source(p, img[b], planes).astype(np.float32)
        w1 = ((stg * np.float32(1 / 255.0) - np.float32(mean)) / np.float32(std)).transpose(2, 0, 1)
        lvl = np.abs(got[b, 1] - w1) * istd255
        jpeg = int(p[A.P_A]) == A.A_JPEG
        hsv = int(p[A.P_B]) in (A.B_HUE_ADD, A.B_MUL_HS, A.B_ADD_HS)
        # the Lab members go through powf / cbrtf twice and an equalisation table in between: a one-level difference of L in a sparse part of
        # the histogram moves the table by several levels, and the way back to RGB amplifies it; CLAHE interpolates 4 tables in float
        labm = int(p[A.P_D]) in (A.D_HISTEQ_LAB, A.D_CLAHE_LAB) or int(p[A.P_B]) == A.B_KMEANS
        clahe = int(p[A.P_D]) == A.D_CLAHE_ALL
        share, worst = (lvl > 0.5).mean(), lvl.max()
        coarse_jpeg = jpeg and p[A.P_A + 1] <= 5          # quality <= 5: every table entry is 255 - one flipped coefficient moves a block by a lot
        chain = sum(int(p[i]) != 0 for i in (A.P_A, A.P_B, A.P_C, A.P_D)) > 1      # a tie in one group is amplified by the next ones
        assert share < (8e-2 if coarse_jpeg else 6e-2 if labm else 3e-2 if chain else 2e-2 if jpeg or hsv or clahe else 1e-2) and \
            worst < (90.0 if coarse_jpeg else 40.0 if jpeg else 25.0 if labm else 12.0 if chain else (3.5 if hsv or clahe else 1.5)), \
            (b, [int(p[i]) for i in (A.P_A, A.P_B, A.P_C, A.P_D)], float(worst), float(share))
        if int(p[A.P_A]) or int(p[A.P_B]) or int(p[A.P_C]) or int(p[A.P_D]):
            assert np.abs(got[b, 1] - got[b, 0]).max() > 1e-3, (b, "the member changed nothing")
    # sample 0: no augmentation -> view 1 == view 0, and view 2 == bilinear warp of view 0's raw pixels (zeros outside)
    np.testing.assert_allclose(got[0, 1], got[0, 0], rtol=0, atol=1e-5)
    raw = torch.from_numpy(img[0].astype(np.float32)).permute(2, 0, 1)[None]
    th = torch.from_numpy(theta[0])[None, :2, :]
    grid = F.affine_grid(th, size=(1, 3, H, W), align_corners=True)         # (size-1) normalisation == align_corners=True
    warped_ref = F.grid_sample(raw, grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0].numpy()
    want2 = (warped_ref * np.float32(1 / 255.0) - np.float32(mean)[:, None, None]) / np.float32(std)[:, None, None]
    np.testing.assert_allclose(got[0, 2], want2, rtol=0, atol=2e-4)
    # every sample: view 2 = the warp of ITS staged image (the restated chain of row [b, 1])
    want = D.augment_views(img, params, theta, mean, std, overlay=planes, warp_maps=maps)
    err = np.abs(got[:, 2] - want[:, 2])
    # (a JPEG member FOLLOWED by a histogram equalisation is a tie amplifier: one DCT coefficient that quantises the other way moves a
    # block by a few levels and the equalisation table stretches that by its slope - whether it happens depends on the image; such rows
    # are held to the median, every other row to the tail)
    amplifier = np.array([int(params[b, 1, A.P_A]) == A.A_JPEG and int(params[b, 1, A.P_D]) in (A.D_HISTEQ_ALL, A.D_HISTEQ_LAB, A.D_CLAHE_ALL,
                                                                                              A.D_CLAHE_LAB) for b in range(B)])
    rest = err[~amplifier]
    assert np.quantile(rest, 0.995) < 2e-2 and (rest > 0.08).mean() < 2e-3, (rest.max(), np.quantile(rest, 0.995), (rest > 0.08).mean())
    for b in np.nonzero(amplifier)[0]:
        assert np.median(err[b]) < 0.1, (b, float(np.median(err[b])))
    # sample 2: identity theta -> view 2 is the augmented image itself
    stg2 = D.staged_source(params[2, 1], img[2], planes).astype(np.float32)
    w2 = ((stg2 * np.float32(1 / 255.0) - np.float32(mean)) / np.float32(std)).transpose(2, 0, 1)
    assert (np.abs(got[2, 2] - w2) * istd255 > 0.5).mean() < 6e-3


def check_seg_to_mask(dev, seed=23):
    """Predicted-mask branch (dino_vision.py:64-66): softmax(seg, dim=1)[:, 1] > 0.5 on fp32 logits - random values,
    exact ties, huge magnitudes, infinities / NaNs (the reference's softmax turns those into NaN -> False)."""
    g = torch.Generator().manual_seed(seed)
    images = 3
    seg = rnd((images + 1, 2, 32, 128), g) * 3.0                   # one image more than asked for: must stay untouched
    flat = seg.view(images + 1, 2, -1)
    flat[0, 1, :512] = flat[0, 0, :512]                            # exact ties -> 0.5 > 0.5 is False
    flat[0, :, 512:1024] *= 1.0e4                                  # large magnitudes: the max is subtracted first
    flat[0, :, 1024:1536] *= 1.0e30
    flat[1, 1, :256] = flat[1, 0, :256] + 2.0 ** -12               # small but decidable margins, both signs
    flat[1, 1, 256:512] = flat[1, 0, 256:512] - 2.0 ** -12
    flat[1, 0, 600] = float("inf"); flat[1, 1, 601] = float("inf"); flat[1, :, 602] = float("inf")
    flat[1, 0, 603] = float("-inf"); flat[1, 1, 604] = float("-inf"); flat[1, :, 605] = float("-inf")
    flat[1, 0, 606] = float("nan"); flat[1, 1, 607] = float("nan")
    want = (F.softmax(seg[:images], dim=1)[:, 1] > 0.5).float()
    got = ops.seg_to_mask(seg.to(dev), images)
    assert got.shape == (images, 32, 128)
    np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())
    # margins below the resolution of exp() near 1 may go either way in any implementation; they must still be a
    # decision between the two classes that agrees with the sign of the margin or with "not greater" (0)
    tiny = rnd((1, 2, 32, 128), g)
    tiny[0, 1] = tiny[0, 0] + (torch.rand(32, 128, generator=g) - 0.5) * 2.0e-7
    got_t = ops.seg_to_mask(tiny.to(dev), 1).cpu()
    sign = (tiny[0, 1] > tiny[0, 0]).float()
    assert bool(((got_t[0] == sign) | (got_t[0] == 0)).all())


def check_predicted_mask_chain(dev):
    """The whole predicted-mask branch on the device, from the REFERENCE's own fp32 segmentation logits (recorded by
    tools/gen_golden.py at epoch 30): softmax > 0.5, connected components, warp to view 2 and row selection reproduce the
    reference's `zero` / `index` bit for bit (dino_vision.py:64-87)."""
    g = np.load(os.path.join(GOLD, "small3_step.npz"))
    step = [k for k in range(4) if g[f"s{k}/hyper"][0] >= 30][0]
    p = f"s{step}/"
    seg = torch.from_numpy(g[p + "seg_logits_view1"])                      # [B,2,32,128]
    B = seg.shape[0]
    mask = ops.seg_to_mask(seg.to(dev), B)
    np.testing.assert_array_equal(mask.cpu().numpy().astype(np.uint8), g[p + "pred_mask"])
    idm1 = ops.ccl_label(mask)
    idm2 = ops.warp_idmap(idm1, torch.from_numpy(g[p + "metrics"]).to(dev))
    both = torch.cat([idm1, idm2])
    np.testing.assert_array_equal(both.cpu().numpy(), g[p + "zero_idmap"])
    assert len(np.unique(g[p + "zero_idmap"])) > 3
    _, _, present = ops.region_stats(both)
    _, _, _, new_index = ops.select_scan(present, B)
    np.testing.assert_array_equal(new_index.cpu().numpy().astype(bool), g[p + "new_index"])


def check_region(dev, E=128, seed=7):
    g = np.load(os.path.join(GOLD, "small_step.npz"))
    ids = np.concatenate([g["s0/zero_idmap"], g["pred/zero_idmap"]])          # two different kinds of maps
    ids = np.concatenate([ids[:8], ids[16:24], ids[8:16], ids[24:32]])         # [view1 x16, view2 x16]
    B = ids.shape[0] // 2
    gen = torch.Generator().manual_seed(seed)
    feat = rnd((2 * B, 256, E), gen).to(BF)
    planes = torch.from_numpy(ccl_np.idmap_to_planes(ids))
    region_f = feat.float().reshape(2 * B, 8, 32, E).permute(0, 3, 1, 2).requires_grad_(True)
    vecs, index = O.region_pool(region_f, planes)
    rows_ref, new_index_ref = O.select_rows(vecs, index)
    idm = torch.from_numpy(ids).to(dev)
    tok_plane, tok_coef, present = ops.region_stats(idm)
    np.testing.assert_array_equal(present.cpu().numpy().astype(bool), index.numpy())
    nsel, offset, total, new_index = ops.select_scan(present, B)
    np.testing.assert_array_equal(new_index.cpu().numpy().astype(bool), new_index_ref.numpy())
    M = int(total.cpu().item())
    assert 2 * M == rows_ref.shape[0]
    rows = torch.zeros((2 * 26 * B, E), dtype=BF).to(dev)
    ops.region_pool_fwd(feat.to(dev), tok_plane, tok_coef, nsel, offset, total, rows, B)
    close(rows[:2 * M], rows_ref, 1e-2, 1e-2, "pool/rows")
    d_rows = torch.zeros((2 * 26 * B, E), dtype=BF)
    d_rows[:2 * M] = rnd((2 * M, E), gen).to(BF)
    rows_ref.backward(d_rows[:2 * M].float())
    d_feat = torch.empty((2 * B, 256, E), dtype=BF).to(dev)
    ops.region_pool_bwd(d_rows.to(dev), tok_plane, tok_coef, nsel, offset, total, d_feat, B)
    want = region_f.grad.permute(0, 2, 3, 1).reshape(2 * B, 256, E)
    close(d_feat, want, 1e-2, 1e-3, "pool/d_feat")
    # dense-plane round trip
    np.testing.assert_array_equal(ops.idmap_to_planes(idm).cpu().numpy(), planes.numpy())
    np.testing.assert_array_equal(ops.planes_to_idmap(planes.to(dev)).cpu().numpy(), ids)


def check_region_adjacent_planes(dev, E=64, seed=17):
    """View 2 of the reference is a warp of 26 separate planes: components that were one pixel apart can end up side by
    side (zoom-out, sub-pixel shift), and a token whose central 2x2 straddles them is credited to BOTH planes
    (dino_vision.py:38-49).  Labelled masks with 1-px gaps + zoom-out / half-pixel thetas, against the dense path."""
    gen = torch.Generator().manual_seed(seed)
    B = 4
    mask = np.zeros((B, 32, 128), dtype=np.float32)
    for b in range(B):
        x = 2 + b
        for k in range(6):                     # 6 characters, 12 wide, separated by ONE background column
            mask[b, 6:26, x:x + 12] = 1.0
            x += 13
    ids1 = ccl_np.label_batch_idmap(mask) if hasattr(ccl_np, "label_batch_idmap") else None
    idm1 = ops.ccl_label(torch.from_numpy(mask).to(dev))
    if ids1 is not None:
        np.testing.assert_array_equal(idm1.cpu().numpy(), ids1)
    theta = torch.tensor([[[1.55, 0.0, 0.0], [0.0, 1.3, 0.0]],          # zoom-out: gaps of the source close
                          [[1.0, 0.0, 1.0 / 128], [0.0, 1.0, 0.0]],      # half-pixel shift
                          [[1.4, 0.12, 0.02], [0.05, 1.2, -0.03]],
                          [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]], dtype=torch.float32)
    planes1 = torch.from_numpy(ccl_np.idmap_to_planes(idm1.cpu().numpy()))
    planes2 = (O.warp_planes(planes1, theta) > 0.1).float()
    idm2 = ops.warp_idmap(idm1, theta.to(dev))
    np.testing.assert_array_equal(ops.idmap_to_planes(idm2).cpu().numpy(), planes2.numpy())     # the id map loses nothing
    # some token of the warped view must really straddle two planes, or this test pins nothing
    c = F.interpolate(planes2, size=(8, 32), mode="bilinear", align_corners=None)
    assert int(((c > 0).sum(1) > 1).sum()) > 0, "no token touches two planes: strengthen the fixture"
    feat = rnd((2 * B, 256, E), gen).to(BF)
    region_f = feat.float().reshape(2 * B, 8, 32, E).permute(0, 3, 1, 2).requires_grad_(True)
    vecs, index = O.region_pool(region_f, torch.cat([planes1, planes2]))
    rows_ref, new_index_ref = O.select_rows(vecs, index)
    tok_plane, tok_coef, present = ops.region_stats(torch.cat([idm1, idm2]))
    np.testing.assert_array_equal(present.cpu().numpy().astype(bool), index.numpy())
    nsel, offset, total, new_index = ops.select_scan(present, B)
    M = int(total.cpu().item())
    assert 2 * M == rows_ref.shape[0]
    rows = torch.zeros((2 * 26 * B, E), dtype=BF).to(dev)
    ops.region_pool_fwd(feat.to(dev), tok_plane, tok_coef, nsel, offset, total, rows, B)
    close(rows[:2 * M], rows_ref, 1e-2, 1e-2, "pool-adjacent/rows")
    d_rows = torch.zeros((2 * 26 * B, E), dtype=BF)
    d_rows[:2 * M] = rnd((2 * M, E), gen).to(BF)
    rows_ref.backward(d_rows[:2 * M].float())
    d_feat = torch.empty((2 * B, 256, E), dtype=BF).to(dev)
    ops.region_pool_bwd(d_rows.to(dev), tok_plane, tok_coef, nsel, offset, total, d_feat, B)
    close(d_feat, region_f.grad.permute(0, 2, 3, 1).reshape(2 * B, 256, E), 1e-2, 1e-3, "pool-adjacent/d_feat")


def check_patch_embed(dev, views=3, E=192, seed=8):
    gen = torch.Generator().manual_seed(seed)
    img = rnd((views, 3, 32, 128), gen)
    w = rnd((E, 3, 4, 4), gen, 0.1).requires_grad_(True)
    b = rnd((E,), gen, 0.1).requires_grad_(True)
    pos = rnd((256, E), gen, 0.1).requires_grad_(True)
    ref = F.conv2d(img, w, b, stride=4).flatten(2).transpose(1, 2) + pos
    out = ops.patch_embed_fwd(img.to(dev), w.detach().to(dev), b.detach().to(dev), pos.detach().to(dev))
    close(out.reshape(views, 256, E), ref, 1e-4, 1e-4, "pe/out")
    gr = rnd((views, 256, E), gen)
    ref.backward(gr)
    dw, db, dp = torch.zeros(E, 48).to(dev), torch.zeros(E).to(dev), torch.zeros(256, E).to(dev)
    ops.patch_embed_bwd(img.to(dev), gr.reshape(-1, E).to(dev), dw, db, dp)
    # d_w / d_bias run on bf16 operands (MFMA product over all tokens): tolerance ~ bf16 eps * sqrt(tokens)
    tol = 1e-2 * (views * 256) ** 0.5
    close(dw.reshape(E, 3, 4, 4), w.grad, 1e-2, tol, "pe/dw")
    close(db, b.grad, 1e-2, tol, "pe/db")
    close(dp, pos.grad, 1e-3, 1e-4, "pe/dpos")
    # the same from a bf16 gradient stream (ccd_patch_embed_bwd_g16: the stream is the product's operand as it lies)
    dw, db, dp = torch.zeros(E, 48).to(dev), torch.zeros(E).to(dev), torch.zeros(256, E).to(dev)
    ops.patch_embed_bwd(img.to(dev), gr.reshape(-1, E).to(BF).to(dev), dw, db, dp)
    close(dw.reshape(E, 3, 4, 4), w.grad, 1e-2, tol, "pe16/dw")
    close(db, b.grad, 1e-2, tol, "pe16/db")
    close(dp, gr.to(BF).float().sum(0), 1e-3, 1e-3, "pe16/dpos")


def check_small_ops(dev, seed=9):
    gen = torch.Generator().manual_seed(seed)
    a, b = rnd((37, 50), gen), rnd((50, 70), gen)
    out = torch.ones(37, 70).to(dev)
    close(ops.small_matmul(a.to(dev), b.to(dev), out, accumulate=True), a @ b + 1, 1e-4, 1e-4, "smm/acc")
    at = rnd((50, 37), gen)
    close(ops.small_matmul(at.to(dev), b.to(dev), torch.empty(37, 70).to(dev), trans_a=True), at.t() @ b, 1e-4, 1e-4,
          "smm/trans")
    x = rnd((333, 136), gen).to(BF)
    out = torch.full((136,), 2.0).to(dev)
    close(ops.colsum_bf16(x.to(dev), out), x.float().sum(0) + 2, 1e-3, 1e-2, "colsum")
    d_rows = torch.tensor([50], dtype=torch.int32).to(dev)
    out = torch.zeros(136).to(dev)
    close(ops.colsum_bf16(x.to(dev), out, d_rows=d_rows, rows_mul=2), x[:100].float().sum(0), 1e-3, 1e-2, "colsum/dyn")
    src = rnd((5000,), gen)
    close(ops.cast_bf16(src.to(dev), torch.empty(5000, dtype=BF).to(dev)), src.to(BF), 0, 0, "cast")
    # mirror (bf16 copy + transposed copy) of two matrices in one launch
    m1, m2 = rnd((70, 45), gen), rnd((33, 130), gen)
    s1, s2 = m1.to(dev), m2.to(dev)
    d1, t1 = torch.empty(70, 45, dtype=BF).to(dev), torch.empty(45, 70, dtype=BF).to(dev)
    t2 = torch.empty(130, 33, dtype=BF).to(dev)
    tiles1 = 2 * 1                  # 64 x 64 tiles: ceil(70 / 64) * ceil(45 / 64); the second matrix has an odd row count (2-byte stores)
    tiles2 = 1 * 3
    blob = struct.pack("PPPiiii", s1.data_ptr(), d1.data_ptr(), t1.data_ptr(), 70, 45, 0, 0)
    blob += struct.pack("PPPiiii", s2.data_ptr(), 0, t2.data_ptr(), 33, 130, tiles1, 0)
    descs = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    ops.mirror_bf16(descs, 2, tiles1 + tiles2)
    close(d1, m1.to(BF), 0, 0, "mirror/d1"); close(t1, m1.t().to(BF), 0, 0, "mirror/t1")
    close(t2, m2.t().to(BF), 0, 0, "mirror/t2")


def check_head_pieces(dev, rows=37, D=64, K=200, seed=10):
    gen = torch.Generator().manual_seed(seed)
    x = rnd((rows, D), gen).to(BF)
    xr = x.float().requires_grad_(True)
    yr = F.normalize(xr, dim=-1, p=2)
    y, inv = torch.empty(rows, D, dtype=BF).to(dev), torch.empty(rows).to(dev)
    d_rows = torch.tensor([rows - 5], dtype=torch.int32).to(dev)
    ops.l2norm_fwd(x.to(dev), y, inv, d_rows=d_rows)
    close(y[:rows - 5], yr[:rows - 5], 1e-2, 1e-2, "l2/y")
    dy = rnd((rows, D), gen).to(BF)
    yr.backward(dy.float())
    dx = torch.zeros(rows, D, dtype=BF).to(dev)
    ops.l2norm_bwd(x.to(dev), inv, dy.to(dev), dx, d_rows=d_rows)
    close(dx[:rows - 5], xr.grad[:rows - 5], 2e-2, 1e-2, "l2/dx")
    v = rnd((K, D), gen).requires_grad_(True)
    gg = (1 + 0.1 * rnd((K, 1), gen)).requires_grad_(True)
    wr = v * (gg / v.norm(dim=1, keepdim=True))
    w, wt, winv = torch.empty(K, D, dtype=BF).to(dev), torch.empty(D, K, dtype=BF).to(dev), torch.empty(K).to(dev)
    ops.weightnorm_fwd(v.detach().to(dev), gg.detach().to(dev), w, wt, winv)
    close(w, wr, 1e-2, 1e-3, "wn/w"); close(wt, wr.t(), 1e-2, 1e-3, "wn/wt")
    dw = rnd((K, D), gen)
    wr.backward(dw)
    dv, dg = torch.empty(K, D).to(dev), torch.empty(K, 1).to(dev)
    ops.weightnorm_bwd(v.detach().to(dev), gg.detach().to(dev), winv, dw.to(dev), dv, dg)
    close(dv, v.grad, 1e-3, 1e-4, "wn/dv"); close(dg, gg.grad, 1e-3, 1e-4, "wn/dg")


def check_dino_loss(dev, M=11, K=4096, seed=11, temp=0.04):
    gen = torch.Generator().manual_seed(seed)
    max_rows = 2 * M + 6
    s = torch.zeros(max_rows, K); t = torch.zeros(max_rows, K)
    s[:2 * M] = rnd((2 * M, K), gen, 0.5); t[:2 * M] = rnd((2 * M, K), gen, 0.5)
    s[3, 77] = 9.0; t[5, 1234] = 6.0                                # force online-softmax rescales late in the row
    center = rnd((1, K), gen, 0.1)
    sr = s[:2 * M].clone().requires_grad_(True)
    ref = O.dino_ce(sr, t[:2 * M], center, temp)
    ref.backward()
    d_m = torch.tensor([M], dtype=torch.int32).to(dev)
    stats = torch.zeros(max_rows, 4).to(dev)
    loss = torch.zeros(1).to(dev)
    S, T_, C_ = s.to(dev), t.to(dev), center.to(dev)
    ops.dino_loss_fwd(S, T_, C_, d_m, 0.1, temp, stats, loss)
    close(loss, ref.reshape(1), 1e-5, 1e-5, "dino/loss")
    dl = torch.zeros(max_rows, K, dtype=BF).to(dev)
    ops.dino_loss_bwd(S, T_, C_, d_m, 0.1, temp, stats, 1.0, dl)
    close(dl[:2 * M], sr.grad, 2e-2, 1e-6, "dino/dlogits")
    bsum = torch.zeros(K).to(dev)
    ops.colsum_f32(T_, bsum, d_rows=d_m, rows_mul=2)
    close(bsum, t[:2 * M].sum(0), 1e-4, 1e-4, "dino/colsum")
    cen = center.clone().reshape(-1).to(dev)
    ops.center_ema(cen, bsum, d_m, 1, 0.9)
    close(cen, O.center_update(center, t[:2 * M]).reshape(-1), 1e-5, 1e-6, "dino/center")


def check_head_loss(dev, M=11, K=512, seed=21, temp=0.04, spare=6):
    """ccd_head_loss_fwd / _bwd (headloss.h): the head's last layer and the distillation loss with the logits in registers, against
    the oracle's loss on logits formed in fp32 from the same bf16 factors, and against the unfused kernels (dino_loss_fwd / _bwd
    on materialised logits).  One weight row of either network is scaled up so that the running maxima move late in a row."""
    gen = torch.Generator().manual_seed(seed)
    D, max_rows = 256, 2 * M + spare
    nrm = lambda x: x / x.norm(dim=1, keepdim=True)
    zs, zt = nrm(rnd((max_rows, D), gen)).to(BF), nrm(rnd((max_rows, D), gen)).to(BF)
    ws, wt = nrm(rnd((K, D), gen)), nrm(rnd((K, D), gen))
    ws[K // 2 + 3] *= 4.0; wt[K - 5] *= 3.0
    ws, wt = ws.to(BF), wt.to(BF)
    center = rnd((1, K), gen, 0.05)
    s = (zs.float() @ ws.float().t())[:2 * M]
    t = (zt.float() @ wt.float().t())[:2 * M]
    sr = s.clone().requires_grad_(True)
    ref = O.dino_ce(sr, t, center, temp)
    ref.backward()
    d_m = torch.tensor([M], dtype=torch.int32).to(dev)
    ZS, ZT, WS, WT, C_ = zs.to(dev), zt.to(dev), ws.to(dev), wt.to(dev), center.reshape(-1).to(dev)
    assert ops.head_loss_supported(K, D, max_rows)
    stats = torch.zeros(max_rows, 4).to(dev)
    loss = torch.zeros(1).to(dev)
    ops.head_loss_fwd(ZS, ZT, WS, WT, C_, d_m, 0.1, temp, stats, loss)
    close(loss, ref.reshape(1), 2e-5, 2e-5, "head_loss/loss")
    dl = torch.full((max_rows, K), 7.0, dtype=BF).to(dev)
    ops.head_loss_bwd(ZS, ZT, WS, WT, C_, d_m, 0.1, temp, stats, 1.0, dl)
    gmax = float(sr.grad.abs().max())
    close(dl[:2 * M], sr.grad, 2e-2, 1e-3 * gmax, "head_loss/dlogits")
    assert bool((dl[2 * M:] == 7.0).all()), "head_loss_bwd wrote behind row 2M"
    # the unfused chain on the same logits (what the fused pair replaces)
    S = torch.zeros(max_rows, K); T_ = torch.zeros(max_rows, K)
    S[:2 * M] = s; T_[:2 * M] = t
    st2, loss2 = torch.zeros(max_rows, 4).to(dev), torch.zeros(1).to(dev)
    ops.dino_loss_fwd(S.to(dev), T_.to(dev), C_, d_m, 0.1, temp, st2, loss2)
    close(loss, loss2, 2e-5, 2e-5, "head_loss/loss vs unfused")
    dl2 = torch.zeros(max_rows, K, dtype=BF).to(dev)
    ops.dino_loss_bwd(S.to(dev), T_.to(dev), C_, d_m, 0.1, temp, st2, 1.0, dl2)
    close(dl[:2 * M], dl2[:2 * M], 2e-2, 1e-3 * gmax, "head_loss/dlogits vs unfused")
    # a device-side upstream gradient and a second call into the same buffers (the workspace is reused)
    gscale = torch.tensor([0.5]).to(dev)
    ops.head_loss_bwd(ZS, ZT, WS, WT, C_, d_m, 0.1, temp, stats, 2.0, dl, d_grad_scale=gscale)
    close(dl[:2 * M], sr.grad, 2e-2, 1e-3 * gmax, "head_loss/dlogits scaled")


def check_seg_loss(dev, half=2, seed=12):
    gen = torch.Generator().manual_seed(seed)
    logits = rnd((2 * half, 2, 32, 128), gen, 2.0)
    ma = (torch.rand((half, 32, 128), generator=gen) > 0.5).float()
    mb = (torch.rand((half, 32, 128), generator=gen) > 0.5)
    lr = logits.clone().requires_grad_(True)
    ref = O.seg_loss(lr, torch.cat([ma, mb.float()]))
    ref.backward()
    idb = torch.where(mb, torch.tensor(0, dtype=torch.uint8), torch.tensor(255, dtype=torch.uint8))
    loss, dlog = torch.zeros(1).to(dev), torch.empty_like(logits).to(dev)
    ops.seg_loss(logits.to(dev), ma.to(dev), idb.to(dev), 1.0, loss, dlog)
    close(loss, ref.reshape(1), 1e-5, 1e-6, "seg/loss")
    close(dlog, lr.grad, 1e-3, 1e-9, "seg/dlogits")


def check_optimizer(dev, seed=13):
    gen = torch.Generator().manual_seed(seed)
    sizes = [5, 1024, 1500, 64, 3000]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    param = rnd((total,), gen); grad = rnd((total,), gen) * torch.tensor(3.0)
    grad[offs[2]:offs[2] + sizes[2]] *= 0.001          # a tensor whose norm stays under the clip
    chunk_seg, chunk_begin, chunk_len = [], [], []
    for s_, (o, n) in enumerate(zip(offs, sizes)):
        for c in range(0, n, 1024):
            chunk_seg.append(s_); chunk_begin.append(o + c); chunk_len.append(min(1024, n - c))
    cs = torch.tensor(chunk_seg, dtype=torch.int32).to(dev)
    cb = torch.tensor(chunk_begin, dtype=torch.int64).to(dev)
    cl = torch.tensor(chunk_len, dtype=torch.int32).to(dev)
    norm2 = torch.zeros(len(sizes)).to(dev)
    G = grad.to(dev)
    ops.seg_sumsq(G, cs, cb, cl, norm2)
    want = torch.stack([grad[o:o + n].pow(2).sum() for o, n in zip(offs, sizes)])
    close(norm2, want, 1e-4, 1e-6, "opt/norm2")
    lr, wd, clip, step = 3e-3, 0.1, 3.0, 4
    active = [1, 1, 1, 0, 1]
    decay = [1, 0, 1, 1, 1]
    hyper = torch.tensor([[lr * wd * d, lr / (1 - 0.9 ** step), 1 / np.sqrt(1 - 0.999 ** step), a]
                          for d, a in zip(decay, active)], dtype=torch.float32).to(dev)
    m0, v0 = rnd((total,), gen) * 0.1, rnd((total,), gen).abs() * 0.1
    Pd, Md, Vd = param.clone().to(dev), m0.clone().to(dev), v0.clone().to(dev)
    mirror = torch.zeros(total, dtype=BF).to(dev)
    ops.adamw(Pd, G, Md, Vd, mirror, cs, cb, cl, hyper, norm2, clip)
    # reference: the oracle's per-tensor clip + AdamW formulas
    for s_, (o, n) in enumerate(zip(offs, sizes)):
        sl = slice(o, o + n)
        if not active[s_]:
            close(Pd[sl], param[sl], 0, 0, "opt/inactive")
            continue
        gsl = grad[sl].clone()
        O.clip_per_tensor({"g": gsl}, clip)
        p = param[sl].clone()
        p.mul_(1 - lr * (wd if decay[s_] else 0.0))
        m = m0[sl] * 0.9 + 0.1 * gsl
        v = v0[sl] * 0.999 + 0.001 * gsl * gsl
        p.addcdiv_(m, (v.sqrt() / np.sqrt(1 - 0.999 ** step)).add_(1e-8), value=-lr / (1 - 0.9 ** step))
        close(Pd[sl], p, 1e-5, 1e-6, f"opt/param{s_}")
        close(Md[sl], m, 1e-5, 1e-7, f"opt/m{s_}")
        close(Vd[sl], v, 1e-5, 1e-7, f"opt/v{s_}")
        close(mirror[sl], p.to(BF), 1e-2, 1e-6, f"opt/mirror{s_}")
    G2 = grad.clone().to(dev)
    ops.clip_scale(G2, cs, cb, cl, norm2, clip)
    for s_, (o, n) in enumerate(zip(offs, sizes)):
        gsl = grad[o:o + n].clone()
        O.clip_per_tensor({"g": gsl}, clip)
        close(G2[o:o + n], gsl, 1e-5, 1e-7, f"opt/clip{s_}")
    teacher = rnd((total + 3,), gen); student = rnd((total + 3,), gen)
    Td, tm = teacher.clone().to(dev), torch.zeros(total + 3, dtype=BF).to(dev)
    ops.ema(Td, student.to(dev), tm, 0.9995)
    close(Td, teacher * 0.9995 + (1 - 0.9995) * student, 1e-6, 1e-7, "opt/ema")
    close(tm, Td.to(BF), 0, 0, "opt/ema-mirror")
    # the momentum read on the device when the kernel runs (replays of a graphed step): the launch arguments are ignored
    Td2, tm2 = teacher.clone().to(dev), torch.zeros(total + 3, dtype=BF).to(dev)
    ops.ema(Td2, student.to(dev), tm2, 0.0, d_m=torch.tensor([0.9995, 1 - 0.9995], dtype=torch.float32).to(dev))
    assert torch.equal(Td2.cpu(), Td.cpu()) and torch.equal(tm2.cpu(), tm.cpu()), "opt/ema device momentum"


def _cos(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def check_conv_pieces(dev, seed=20):
    """Implicit-GEMM conv / transposed-conv parity classes / im2col / BN+ReLU / classifier conv vs torch.nn.functional."""
    from ccd_amd import seghead as sh
    g = torch.Generator().manual_seed(seed)
    n, gh, gw, cin, cout = 2, 4, 8, 64, 16
    x = rnd((n, cin, gh, gw), g).to(BF)
    w = rnd((cout, cin, 3, 3), g, 0.1)
    rows = n * gh * gw
    xr = x.permute(0, 2, 3, 1).reshape(rows, cin).contiguous().to(dev)
    wq = w.to(BF).float()
    # --- 3x3 forward with statistics
    wb = ops.permute4(w.to(dev), (cin * 9, 1, 9), (cout, 9, cin), torch.empty((cout, 9 * cin), dtype=BF, device=dev))
    d3 = ops.conv_desc((gh, gw), (gh, gw), cin, sh.TAPS3)
    stats = torch.zeros(2 * cout, device=dev)
    y = ops.conv_gemm(xr, d3, wb, rows, torch.empty((rows, cout), dtype=BF, device=dev), colsum=stats[:cout],
                      colsumsq=stats[cout:])
    ref = F.conv2d(x.float(), wq, padding=1).permute(0, 2, 3, 1).reshape(rows, cout)
    close(y, ref, 1e-2, 2e-2, "conv3x3")
    close(stats[:cout], ref.sum(0), 1e-3, 0.1, "conv3x3/colsum")
    close(stats[cout:], (ref * ref).sum(0), 1e-2, 0.1, "conv3x3/colsumsq")
    # --- im2col
    cols = ops.im2col(xr, d3, rows)
    ref_cols = F.unfold(x.float(), 3, padding=1).view(n, cin, 9, gh * gw).permute(0, 3, 2, 1).reshape(rows, 9 * cin)
    assert torch.equal(cols.float().cpu(), ref_cols), "im2col"
    # --- weight gradient with the implicit patch matrix == TN GEMM against the explicit one
    gy = rnd((rows, 24), g).to(BF).to(dev)
    wg_impl = ops.conv_wgrad(gy, xr, d3, torch.zeros((24, 9 * cin), device=dev))
    want_wg = gy.float().cpu().t() @ ref_cols
    close(wg_impl, want_wg, 1e-3, 1e-2, "conv3x3/wgrad")
    # --- 3x3 data gradient: dX = conv_transpose(dY, W)
    dy = rnd((n, cout * 4, gh, gw), g).to(BF)                    # 64 channels (cin of the gradient gather)
    w2 = rnd((cout * 4, cin, 3, 3), g, 0.1)
    dyr = dy.permute(0, 2, 3, 1).reshape(rows, cout * 4).contiguous().to(dev)
    wd = ops.permute4(w2.to(dev), (9, 1, cin * 9), (cin, 9, cout * 4),
                      torch.empty((cin, 9 * cout * 4), dtype=BF, device=dev))
    d3f = ops.conv_desc((gh, gw), (gh, gw), cout * 4, sh.TAPS3_FLIP)
    dx = ops.conv_gemm(dyr, d3f, wd, rows, torch.empty((rows, cin), dtype=BF, device=dev))
    ref = F.conv_transpose2d(dy.float(), w2.to(BF).float(), padding=1).permute(0, 2, 3, 1).reshape(rows, cin)
    close(dx, ref, 1e-2, 3e-2, "conv3x3/dgrad")
    # --- transposed conv 4x4 stride 2: four parity classes + bias
    wt = rnd((cin, cout, 4, 4), g, 0.1)
    bias = rnd((cout,), g)
    up = torch.zeros((4 * rows, cout), dtype=BF, device=dev)
    for py in (0, 1):
        for px in (0, 1):
            pt = sh._parity_taps(py, px)
            desc = ops.conv_desc((gh, gw), (gh, gw), cin, [(a, b) for _, _, a, b in pt], parity=(py, px))
            wp = torch.empty((cout, 4 * cin), dtype=BF, device=dev)
            ops.permute4(wt.to(dev).reshape(-1)[pt[0][0] * 4 + pt[0][1]:], (16, 8, 2, cout * 16), (cout, 2, 2, cin), wp)
            ops.conv_gemm(xr, desc, wp, rows, up, bias=bias.to(dev))
    ref = F.conv_transpose2d(x.float(), wt.to(BF).float(), bias, stride=2, padding=1)
    close(up, ref.permute(0, 2, 3, 1).reshape(4 * rows, cout), 1e-2, 3e-2, "convT")
    # --- transposed conv data gradient: dIn = conv2d(dOut, W^T-ish, stride 2)
    dout = rnd((n, 64, 2 * gh, 2 * gw), g).to(BF)
    wt2 = rnd((cin, 64, 4, 4), g, 0.1)
    dor = dout.permute(0, 2, 3, 1).reshape(4 * rows, 64).contiguous().to(dev)
    descg = ops.conv_desc((gh, gw), (2 * gh, 2 * gw), 64, sh.TAPS_T_GRAD, s_mul=2)
    wg = ops.permute4(wt2.to(dev), (64 * 16, 1, 16), (cin, 16, 64), torch.empty((cin, 16 * 64), dtype=BF, device=dev))
    din = ops.conv_gemm(dor, descg, wg, rows, torch.empty((rows, cin), dtype=BF, device=dev))
    ref = F.conv2d(dout.float(), wt2.to(BF).float(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(rows, cin)
    close(din, ref, 1e-2, 5e-2, "convT/dgrad")
    # transposed-conv weight gradient: dW[ci][tap][co] = sum_p x[p,ci] * dOut[gather(p,tap),co]
    wg_t = ops.conv_wgrad(xr, dor, descg, torch.zeros((cin, 16 * 64), device=dev))
    want = xr.float().cpu().t() @ ops.im2col(dor, descg, rows).float().cpu()
    close(wg_t, want, 1e-3, 2e-2, "convT/wgrad")
    # --- BatchNorm + ReLU forward / backward
    C = 16
    xb = (rnd((rows, C), g) * 2 + 0.5).to(BF)
    gamma, beta = rnd((C,), g).abs() + 0.5, rnd((C,), g) * 0.3
    rm, rv = torch.zeros(C).to(dev), torch.ones(C).to(dev)
    st = torch.cat([xb.float().sum(0), (xb.float() ** 2).sum(0)]).to(dev)
    mr = torch.empty(2 * C, device=dev)
    ops.bn_finalize(st, rows, 1e-5, 0.1, mr, rm, rv)
    xt = xb.float().clone().requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = torch.zeros(C), torch.ones(C)
    yt = F.relu(F.batch_norm(xt, rm_ref, rv_ref, gt, bt, True, 0.1, 1e-5))
    close(rm, rm_ref, 1e-4, 1e-5, "bn/running_mean"); close(rv, rv_ref, 1e-4, 1e-5, "bn/running_var")
    yb = ops.bn_relu_fwd(xb.to(dev), mr, gamma.to(dev), beta.to(dev), torch.empty((rows, C), dtype=BF, device=dev))
    close(yb, yt, 1e-2, 1e-2, "bn/fwd")
    dyb = rnd((rows, C), g).to(BF)
    yt.backward(dyb.float())
    red = torch.zeros(2 * C, device=dev)
    ops.bn_relu_bwd_reduce(dyb.to(dev), xb.to(dev), mr, gamma.to(dev), beta.to(dev), red)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dxb = ops.bn_relu_bwd_apply(dyb.to(dev), xb.to(dev), mr, gamma.to(dev), beta.to(dev), red, rows, red, dgam, dbet,
                                torch.empty((rows, C), dtype=BF, device=dev))
    close(dxb, xt.grad, 1e-2, 1e-2, "bn/dx"); close(dgam, gt.grad, 1e-3, 1e-3, "bn/dgamma")
    close(dbet, bt.grad, 1e-3, 1e-3, "bn/dbeta")
    # --- classifier conv
    Cc, H, W = 64, 8, 16
    xc = rnd((n, Cc, H, W), g).to(BF)
    wc, bc = rnd((2, Cc, 3, 3), g, 0.2), rnd((2,), g)
    xcr = xc.permute(0, 2, 3, 1).reshape(n * H * W, Cc).contiguous().to(dev)
    lg = sh.cls_forward(xcr, wc.to(dev), bc.to(dev), n, H, W)
    xct = xc.float().requires_grad_(True)
    wct, bct = wc.clone().requires_grad_(True), bc.clone().requires_grad_(True)
    wct_q = wct.to(BF).float()                       # the GEMM reads bf16-rounded weights; gradient w.r.t. those
    wct_q.retain_grad()
    ref = F.conv2d(xct, wct_q, bct, padding=1)
    close(lg, ref, 1e-4, 1e-3, "cls/fwd")
    dl = rnd((n, 2, H, W), g)
    ref.backward(dl)
    dw, db = torch.zeros_like(wc).to(dev), torch.zeros(2).to(dev)
    dxc = sh.cls_backward(dl.to(dev), xcr, wc.to(dev), dw, db, n, H, W)
    close(dxc, xct.grad.permute(0, 2, 3, 1).reshape(n * H * W, Cc), 1e-2, 1e-2, "cls/dx")
    close(dw, wct_q.grad, 1e-2, 0.2, "cls/dw"); close(db, bct.grad, 1e-2, 0.2, "cls/db")


def check_multi_launch_helpers(dev, seed=63):
    """ccd_permute4_multi / ccd_bn_finalize_multi == their one-job-per-launch originals, bit for bit."""
    g = torch.Generator().manual_seed(seed)
    srcs = [rnd((5, 7, 3, 4), g), rnd((16, 9, 8), g), rnd((300,), g), rnd((2, 6, 3, 3), g)]
    specs = [((84, 4, 1, 12), (5, 3, 4, 7)), ((1, 8, 72), (8, 9, 16)), ((1,), (300,)), ((9, 54, 1), (6, 2, 9))]
    jobs, want = [], []
    for src, (strides, dims) in zip(srcs, specs):
        n = 1
        for d in dims:
            n *= d
        sd = src.to(dev)
        want.append(ops.permute4(sd, strides, dims, torch.empty(n, dtype=BF, device=dev)))
        jobs.append((sd, strides, dims, torch.full((n,), 9.0, dtype=BF, device=dev)))
    ops.permute4_multi(jobs)
    for j, w_ in zip(jobs, want):
        assert torch.equal(j[3].cpu().view(torch.int16), w_.cpu().view(torch.int16)), "permute4_multi"
    accs = [torch.full((n_,), 2.0, device=dev) for n_ in (420, 1152)]
    acc_jobs = [(srcs[0].to(dev), specs[0][0], specs[0][1], accs[0]), (srcs[1].to(dev), specs[1][0], specs[1][1], accs[1])]
    ops.permute4_multi(acc_jobs, accumulate=True)
    for (src, strides, dims, got) in acc_jobs:
        ref = ops.permute4(src, strides, dims, torch.full_like(got, 2.0), accumulate=True)
        assert torch.equal(got.cpu(), ref.cpu()), "permute4_multi/accumulate"
    more = [(srcs[2].to(dev), (1,), (300,), torch.empty(300, dtype=BF, device=dev)) for _ in range(30)]      # more jobs than one launch holds
    ops.permute4_multi(more)
    assert all(torch.equal(m[3].cpu().view(torch.int16), want[2].cpu().view(torch.int16)) for m in more)
    # BatchNorm finalisation of three layers of different widths at once
    single, multi = [], []
    for C in (128, 64, 40):
        stats = torch.cat([rnd((C,), g) * 50, torch.rand((C,), generator=g) * 900 + 400]).to(dev)
        rm, rv = rnd((C,), g).to(dev), (torch.rand((C,), generator=g) + 0.5).to(dev)
        a = (stats, 1000.0, 1e-5, 0.1, torch.empty(2 * C, device=dev), rm.clone(), rv.clone())
        ops.bn_finalize(*a)
        single.append(a)
        multi.append((stats, 1000.0, 1e-5, 0.1, torch.empty(2 * C, device=dev), rm.clone(), rv.clone(),
                      torch.full((), 41, dtype=torch.int64, device=dev)))
    ops.bn_finalize_multi(multi)
    for a, b in zip(single, multi):
        for i in (4, 5, 6):
            assert torch.equal(a[i].cpu(), b[i].cpu()), "bn_finalize_multi"
        assert int(b[7]) == 42


def check_cls_tail(dev, images=2, seed=61, ld_pad=0):
    """The fused tail of the segmentation head (kernels/cls_tail.h: BatchNorm + ReLU + Conv2d(128, 2, 3) forward; the classifier's
    data gradient + BatchNorm's backward sums; dy, the transposed conv's bias gradient, cls.weight's gradient) against torch
    (F.conv2d + autograd in fp64 on the operands the kernels see: a and dlogits rounded to bf16, w rounded to bf16) and against
    the unfused kernels of conv.h it replaces."""
    from ccd_amd import seghead as sh
    g = torch.Generator().manual_seed(seed)
    C, H, W = 128, 32, 128
    P = images * H * W
    y_full = torch.empty((P, C + ld_pad), dtype=BF)
    y_full[:, :C] = (rnd((P, C), g) * 1.3 + 0.2).to(BF)
    y_full[:, C:] = 77.0                                                      # (a padded row pitch: never read)
    y = y_full[:, :C]
    yf = y.float()
    mean, var = yf.mean(0) + 0.05 * rnd((C,), g), yf.var(0, unbiased=False) * (1.0 + 0.1 * torch.rand((C,), generator=g))
    mean_rstd = torch.cat([mean, torch.rsqrt(var + 1e-5)]).contiguous()
    gamma, beta = torch.empty(C).uniform_(0.5, 1.5, generator=g), torch.empty(C).uniform_(-0.3, 0.3, generator=g)
    w, bias = rnd((2, C, 3, 3), g, 0.05), rnd((2,), g, 0.1)
    dl = rnd((images, 2, H, W), g) / 64.0
    dl[0, :, 0, :5] *= 40.0                                                   # corners / edges carry weight (the zero padding)
    dl[-1, :, H - 1, W - 3:] *= 40.0
    yd = y_full.to(dev)[:, :C]
    mr_d, ga_d, be_d, w_d, b_d, dl_d = (t.to(dev) for t in (mean_rstd, gamma, beta, w, bias, dl))
    assert ops.cls_tail_supported(yd, H, W)
    # ---- forward
    got = ops.cls_tail_fwd(yd, mr_d, ga_d, be_d, w_d, b_d, images, H, W)
    xh = (yf - mean) * mean_rstd[C:]
    pre = xh * gamma + beta
    a = torch.relu(pre).to(BF).double()                                       # the operand of the product
    wq = w.to(BF).double()
    a_img = a.view(images, H, W, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wq.requires_grad_(True)
    want = F.conv2d(a_img, wq, bias.double(), padding=1)
    close(got, want, 2e-4, 2e-4, "cls_tail/logits")
    a_dev = ops.bn_relu_fwd(yd, mr_d, ga_d, be_d, torch.empty((P, C), dtype=BF, device=dev))
    unfused = sh.cls_forward(a_dev, w_d, b_d, images, H, W)
    close(got, unfused, 1e-5, 2e-5, "cls_tail/logits vs conv.h")
    # ---- backward: the sums
    dlq = dl.to(BF).double()
    want.backward(dlq)
    d_a = a_img.grad.permute(0, 2, 3, 1).reshape(P, C)
    mask = (pre > 0).double()
    d = d_a * mask
    red_want = torch.cat([d.sum(0), (d * xh.double()).sum(0)])
    red = torch.zeros(2 * C, device=dev)
    db = torch.full((2,), 0.25, device=dev)
    ops.cls_tail_bwd_reduce(dl_d, yd, mr_d, ga_d, be_d, w_d, red, db, images, H, W)
    scale = float(red_want.abs().max())
    close(red, red_want, 2e-3, 2e-3 * scale, "cls_tail/red")
    close(db, 0.25 + dl.double().sum((0, 2, 3)), 1e-4, 1e-5, "cls_tail/db_cls")
    # ---- backward: dy and the parameter gradients (red as another rank would have changed it: not this rank's own sums)
    red_all = (red_want * 1.7).float()
    count = float(P) * 1.7
    k0, k1 = red_all[:C].double() / count, red_all[C:].double() / count
    dy_want = (gamma * mean_rstd[C:]).double() * (d - k0 - xh.double() * k1)
    dy_full = torch.full((P, C + ld_pad), 5.0, dtype=BF, device=dev)
    dgamma, dbeta = torch.full((C,), 1.0, device=dev), torch.full((C,), -2.0, device=dev)
    dw, dbt = torch.full((2, C, 3, 3), 0.5, device=dev), torch.full((C,), 3.0, device=dev)
    red_local = torch.cat([d.sum(0), (d * xh.double()).sum(0)]).float().to(dev)
    ops.cls_tail_bwd_apply(dl_d, yd, mr_d, ga_d, be_d, w_d, red_all.to(dev), count, red_local, dgamma, dbeta, dw, dbt, dy_full[:, :C],
                           images, H, W)
    close(dy_full[:, :C], dy_want, 1.2e-2, 1e-2 * float(dy_want.abs().max()) / 50, "cls_tail/dy")
    if ld_pad:
        assert bool((dy_full[:, C:].float() == 5.0).all()), "cls_tail/dy wrote past its 128 columns"
    close(dbeta, -2.0 + red_local[:C], 1e-6, 1e-6, "cls_tail/dbeta")
    close(dgamma, 1.0 + red_local[C:], 1e-6, 1e-6, "cls_tail/dgamma")
    close(dbt, 3.0 + dy_full[:, :C].double().sum(0), 1e-3, 1e-3 * float(dy_want.abs().max()) * math.sqrt(P), "cls_tail/dbias_t")
    close(dw, 0.5 + wq.grad, 2e-3, 2e-3 * float(wq.grad.abs().max()), "cls_tail/dw_cls")
    # ---- and against the unfused chain on the same inputs
    dx_u = sh.cls_backward(dl_d, a_dev, w_d, torch.zeros((2, C, 3, 3), device=dev), torch.zeros(2, device=dev), images, H, W)
    red_u = torch.zeros(2 * C, device=dev)
    ops.bn_relu_bwd_reduce(dx_u, yd, mr_d, ga_d, be_d, red_u)
    close(red, red_u, 2e-2, 2e-2 * scale, "cls_tail/red vs conv.h")            # (conv.h rounds d(a) to bf16 on the way)


def check_seghead(dev, images=1, E=64, seed=21, build_ref=None):
    """SegHeadFn (HIP) vs the reference-shaped nn.Module stack in fp32 (torch library convs) on the same weights."""
    import copy
    from ccd_amd import seghead as sh
    from ccd_amd.modules.segmentor import SegHead
    torch.manual_seed(seed)
    head = SegHead(in_channels=E)
    with torch.no_grad():                       # non-trivial BN affine parameters
        for m in head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(head).float()
    head = head.to(dev)
    g = torch.Generator().manual_seed(seed)
    taps = [rnd((images * 256, E), g).to(BF) for _ in range(3)]
    tin = [t.to(dev).requires_grad_(True) for t in taps]
    logits = sh.seg_head_forward(head, tin, images)
    assert all(int(m.num_batches_tracked) == 1 for n_, m in head.named_modules()
               if isinstance(m, torch.nn.BatchNorm2d) and not n_.startswith("conv_mla")), "num_batches_tracked"
    rin = [t.float().view(images, 8, 32, E).permute(0, 3, 1, 2).contiguous().detach().requires_grad_(True) for t in taps]
    x = ref.mlahead(*rin)
    want = ref.cls(ref.unpool2(ref.unpool1(x)))
    close(logits, want, 3e-2, 3e-2, "seghead/logits")
    dl = rnd(tuple(want.shape), g) / want[0].numel()
    want.backward(dl)
    logits.backward(dl.to(dev))
    for i in range(3):
        wg = rin[i].grad.permute(0, 2, 3, 1).reshape(images * 256, E)
        c = _cos(tin[i].grad, wg)
        print(f"d_tap{i}: cos {c:.5f}")
        assert c > 0.99, f"seghead/d_tap{i}: cos {c}"
        ratio = float(tin[i].grad.float().norm() / wg.norm())
        assert 0.97 < ratio < 1.03, f"seghead/d_tap{i}: norm ratio {ratio}"
    refp = dict(ref.named_parameters())
    worst = (1.0, None)
    for name, p in head.named_parameters():
        if name.startswith("conv_mla"):
            continue
        gr = refp[name].grad
        # ConvTranspose / 1x1-conv biases feeding a BatchNorm have an exactly-zero true gradient: compare absolutely
        if name in ("unpool1.0.bias", "unpool2.0.bias"):
            scale = float(refp[name.replace("0.bias", "0.weight")].grad.abs().max())
            assert float(p.grad.abs().max()) < 0.05 * scale + 1e-6, f"{name}: {p.grad.abs().max()} vs scale {scale}"
            continue
        c = _cos(p.grad, gr)
        print(f"{name}: cos {c:.5f} ratio {float(p.grad.float().norm() / gr.norm()):.4f}")
        if c < worst[0]:
            worst = (c, name)
        assert c > 0.99, f"seghead/{name}: cos {c}"
        ratio = float(p.grad.float().norm() / gr.norm())
        assert 0.95 < ratio < 1.05, f"seghead/{name}: norm ratio {ratio}"
    refb = dict(ref.named_buffers())
    for name, b in head.named_buffers():
        if name.startswith("conv_mla") or "num_batches" in name:
            continue
        close(b, refb[name], 2e-2, 2e-3, f"seghead/{name}")
    return worst


def check_gemm_resid_ln(dev, M, N, K, seed=30):
    """Residual product + folded LayerNorm == gemm_nt(EPI_RESID) followed by ln_fwd (and the plain fp32 math)."""
    g = torch.Generator().manual_seed(seed)
    a = rnd((M, K), g).to(BF); b = rnd((N, K), g, 0.2).to(BF)
    bias, resid = rnd((N,), g), rnd((M, N), g) * 3 + 0.5
    rps = 16
    rowscale = (torch.rand((M + rps - 1) // rps, generator=g) > 0.3).float() * 1.25
    gamma, beta = rnd((N,), g).abs() + 0.5, rnd((N,), g) * 0.3
    out, y, mean, rstd = ops.gemm_nt_resid_ln(a.to(dev), b.to(dev), bias=bias.to(dev), resid=resid.to(dev),
                                              rowscale=rowscale.to(dev), rows_per_sample=rps, gamma=gamma.to(dev),
                                              beta=beta.to(dev), eps=1e-6)
    want = resid + (a.float() @ b.float().t() + bias) * rowscale.repeat_interleave(rps)[:M, None]
    close(out, want, 1e-4, 1e-4, "resid_ln/out")
    mu, var = want.mean(1), want.var(1, unbiased=False)
    close(mean, mu, 1e-4, 1e-4, "resid_ln/mean")
    close(rstd, (var + 1e-6).rsqrt(), 1e-3, 1e-4, "resid_ln/rstd")
    close(y, F.layer_norm(want, (N,), gamma, beta, 1e-6), 1e-2, 2e-2, "resid_ln/y")


def check_mlp_fused(dev, M, E, H, rps=128, seed=31, store_u=True):
    """One launch == fc1 + GELU + fc2 + residual (DropPath scale) + LayerNorm, in plain fp32 math on the same
    bf16-rounded operands (the hidden activation is rounded to bf16 twice, as the unfused kernels do)."""
    g = torch.Generator().manual_seed(seed)
    y = rnd((M, E), g).to(BF); w1 = rnd((H, E), g, 0.08).to(BF); w2 = rnd((E, H), g, 0.05).to(BF)
    b1, b2 = rnd((H,), g) * 0.5, rnd((E,), g)
    resid = rnd((M, E), g) * 3 + 0.5
    rowscale = (torch.rand((M + rps - 1) // rps, generator=g) > 0.3).float() * 1.25
    if rowscale.numel() > 1:
        rowscale[1] = 0.0                                  # a dropped sample is always present
    gamma, beta = rnd((E,), g).abs() + 0.5, rnd((E,), g) * 0.3
    for rs in (rowscale, None):
        out, yn, mean, rstd, u, gact = ops.mlp_fused(y.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), resid=resid.to(dev),
                                                     rowscale=None if rs is None else rs.to(dev), rows_per_sample=rps,
                                                     gamma=gamma.to(dev), beta=beta.to(dev), eps=1e-6, store_u=store_u,
                                                     store_gact=True)
        assert (gact is not None) == store_u
        u_ref = (y.float() @ w1.float().t() + b1).to(BF)
        h_ref = F.gelu(u_ref.float()).to(BF).float()
        scale = 1.0 if rs is None else rs.repeat_interleave(rps)[:M, None]
        want = resid + (h_ref @ w2.float().t() + b2) * scale
        tag = "mlp_fused" + ("" if rs is not None else "/noscale")
        if store_u:
            # bf16 rounding of an fp32 sum taken in a different order: allow one bf16 ulp on a few elements;
            # row tiles that consist of dropped samples only skip the products and store u = 0 (finite: the
            # backward pass multiplies a zero gradient by gelu'(u) there)
            u_want = u_ref.clone()
            if rs is not None and rps % 128 == 0:
                for t0 in range(0, M, 128):
                    if rs[t0 // rps] == 0:
                        u_want[t0:t0 + 128] = 0
            close(u, u_want, 8e-3, 1e-3, tag + "/u")
            # gelu(u) of the STORED u (what the backward's gelu'(u) product would derive from it), exactly the second product's operand
            close(gact, F.gelu(u.float().cpu()), 8e-3, 1e-6, tag + "/gelu(u)")
            plain = ops.mlp_fused(y.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), resid=resid.to(dev),
                                  rowscale=None if rs is None else rs.to(dev), rows_per_sample=rps, gamma=gamma.to(dev),
                                  beta=beta.to(dev), eps=1e-6, store_u=True)
            assert len(plain) == 5 and torch.equal(plain[0], out) and torch.equal(plain[4], u)
        # bf16 roundings of u / gelu(u) that flip by one ulp move a row sum by ~2e-4 each: tolerance grows with sqrt(H)
        close(out, want, 2e-3, 3e-3 * max(1.0, (H / 128) ** 0.5), tag + "/out")
        mu, var = want.mean(1), want.var(1, unbiased=False)
        close(mean, mu, 1e-3, 1e-3, tag + "/mean")
        close(rstd, (var + 1e-6).rsqrt(), 2e-3, 1e-4, tag + "/rstd")
        close(yn, F.layer_norm(want, (E,), gamma, beta, 1e-6), 1e-2, 2e-2, tag + "/y")
        if rs is not None and rs.numel() > 1:
            lo, hi = rps, min(2 * rps, M)
            assert torch.equal(out[lo:hi].cpu(), resid[lo:hi]), "dropped sample: the stream must pass through unchanged"


def check_proj_mlp_fused(dev, M, E, H, rps=128, seed=41, save=True, drops=True):
    """ccd_proj_mlp_fused == proj + residual (DropPath) + LayerNorm-2 + fc1 + GELU + fc2 + residual (DropPath) + LayerNorm in plain
    fp32 math on the same bf16-rounded operands (y2, u and gelu(u) rounded to bf16 where the kernels round them), i.e.
    == ops.gemm_nt_resid_ln followed by ops.mlp_fused.  drops: samples with either branch, both or none dropped."""
    g = torch.Generator().manual_seed(seed)
    a = rnd((M, E), g).to(BF); wp = rnd((E, E), g, 0.08).to(BF)
    w1 = rnd((H, E), g, 0.08).to(BF); w2 = rnd((E, H), g, 0.05).to(BF)
    bp, b1, b2 = rnd((E,), g), rnd((H,), g) * 0.5, rnd((E,), g)
    resid = rnd((M, E), g) * 3 + 0.5
    ns = (M + rps - 1) // rps
    rs1 = (torch.rand(ns, generator=g) > 0.3).float() * 1.25
    rs2 = (torch.rand(ns, generator=g) > 0.3).float() * 1.25
    pattern = [(1.25, 1.25), (0.0, 1.25), (1.25, 0.0), (0.0, 0.0), (1.25, 1.25)]      # every combination is present when M allows
    for i in range(min(ns, len(pattern))):
        rs1[i], rs2[i] = pattern[i]
    ga2, be2 = rnd((E,), g).abs() + 0.5, rnd((E,), g) * 0.3
    ga, be = rnd((E,), g).abs() + 0.5, rnd((E,), g) * 0.3
    gat, bet = rnd((E,), g).abs() + 0.5, rnd((E,), g) * 0.3
    for use in ((True, False) if drops else (False,)):
        r1, r2 = (rs1, rs2 if save else None) if use else (None, None)      # (a dropped MLP branch reads x_mid back: only with save)
        tap_kw = dict(tap_gamma=gat.to(dev), tap_beta=bet.to(dev)) if use == save else {}     # (with and without the tap output)
        out, yn, mean, rstd, saved, *tap = ops.proj_mlp_fused(
            a.to(dev), wp.to(dev), bp.to(dev), resid=resid.to(dev), rowscale1=None if r1 is None else r1.to(dev), gamma2=ga2.to(dev),
            beta2=be2.to(dev), w1=w1.to(dev), b1=b1.to(dev), w2=w2.to(dev), b2=b2.to(dev), rowscale2=None if r2 is None else r2.to(dev),
            rows_per_sample=rps, gamma=ga.to(dev), beta=be.to(dev), eps=1e-6, save=save, **tap_kw)
        assert len(tap) == (1 if tap_kw else 0)
        s1 = 1.0 if r1 is None else r1.repeat_interleave(rps)[:M, None]
        s2 = 1.0 if r2 is None else r2.repeat_interleave(rps)[:M, None]
        xmid_ref = resid + (a.float() @ wp.float().t() + bp) * s1
        y2_ref = F.layer_norm(xmid_ref, (E,), ga2, be2, 1e-6).to(BF)
        u_ref = (y2_ref.float() @ w1.float().t() + b1).to(BF)
        h_ref = F.gelu(u_ref.float()).to(BF).float()
        want = xmid_ref + (h_ref @ w2.float().t() + b2) * s2
        tag = "proj_mlp_fused" + ("" if use else "/noscale") + ("" if save else "/nosave")
        if save:
            xmid, y2, mean2, rstd2, u = saved
            close(xmid, xmid_ref, 1e-4, 2e-4, tag + "/x_mid")
            close(mean2, xmid_ref.mean(1), 1e-3, 1e-3, tag + "/mean2")
            close(rstd2, (xmid_ref.var(1, unbiased=False) + 1e-6).rsqrt(), 2e-3, 1e-4, tag + "/rstd2")
            close(y2, y2_ref, 1e-2, 2e-2, tag + "/y2")
            # u against the kernel's OWN y2 to a bf16 ulp (also under a dropped MLP branch: its products run and are discarded); against
            # the reference's y2 only loosely - y2 elements that round the other way (LayerNorm arithmetic in another order) move a
            # 384-long sum by ~1e-3 each, 0.03 at the tail of 40 000 elements
            close(u, (y2.float().cpu() @ w1.float().t() + b1).to(BF), 8e-3, 1e-3, tag + "/u (from the stored y2)")
            close(u, u_ref, 1.6e-2, 8e-2, tag + "/u")
        else:
            assert saved is None
        # Against a chain with an INDEPENDENT LayerNorm-2: y2 elements that round the other way (a few per cent of them, the sum
        # order of the statistics differs) move a hidden row, and through gelu' . w2 the output by ~4e-4 each - 7e-3 at the tail of
        # 10^5 elements.  The tight comparison is the one on the kernel's own y2 (save=True) below.
        loose = 1.2e-2 * max(1.0, (H / 128) ** 0.5)
        close(out, want, 2e-3, loose, tag + "/out")
        if save:
            xm_k, y2_k = saved[0].float().cpu(), saved[1].float().cpu()
            u_k = (y2_k @ w1.float().t() + b1).to(BF)
            h_k = F.gelu(u_k.float()).to(BF).float()
            close(out, xm_k + (h_k @ w2.float().t() + b2) * s2, 2e-3, 3e-3 * max(1.0, (H / 128) ** 0.5), tag + "/out (from the stored x_mid, y2)")
        mu, var = want.mean(1), want.var(1, unbiased=False)
        close(mean, mu, 1e-3, 1e-3, tag + "/mean")
        close(rstd, (var + 1e-6).rsqrt(), 2e-3, 1e-4, tag + "/rstd")
        close(yn, F.layer_norm(want, (E,), ga, be, 1e-6), 1e-2, 3e-2, tag + "/y")
        if tap:
            close(tap[0], F.layer_norm(want, (E,), gat, bet, 1e-6), 1e-2, 3e-2, tag + "/tap")
            close(tap[0], F.layer_norm(out.float().cpu(), (E,), gat, bet, 1e-6), 1e-2, 1e-2, tag + "/tap (of the kernel's own rows)")
        if use:
            for i in range(min(ns, len(pattern))):
                lo, hi = i * rps, min((i + 1) * rps, M)
                if rs1[i] == 0 and rs2[i] == 0 and save:
                    assert torch.equal(out[lo:hi].cpu(), resid[lo:hi]), "both branches dropped: the stream must pass through unchanged"
                if rs2[i] == 0 and save:
                    assert torch.equal(out[lo:hi].cpu(), saved[0][lo:hi].cpu()), "dropped MLP branch: x_out must equal x_mid"
        # the same rows through the two separate launches it replaces
        o2, y2b, m2b, r2b = ops.gemm_nt_resid_ln(a.to(dev), wp.to(dev), bias=bp.to(dev), resid=resid.to(dev),
                                                 rowscale=None if r1 is None else r1.to(dev), rows_per_sample=rps, gamma=ga2.to(dev),
                                                 beta=be2.to(dev), eps=1e-6)
        o3 = ops.mlp_fused(y2b, w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), resid=o2, rowscale=None if r2 is None else r2.to(dev),
                           rows_per_sample=rps, gamma=ga.to(dev), beta=be.to(dev), eps=1e-6, store_u=False)
        close(out, o3[0].float().cpu(), 2e-3, loose, tag + "/out vs the two launches")


def check_matvec_bf16(dev, K=1000, D=256, seed=51):
    """out += w . v, and what it is for: the column sums of logits = zn @ w^T without the logits (the teacher centre)."""
    g = torch.Generator().manual_seed(seed)
    w = rnd((K, D), g, 0.1).to(BF)
    v = rnd((D,), g)
    out0 = rnd((K,), g)
    out = ops.matvec_bf16(w.to(dev), v.to(dev), out0.clone().to(dev))
    close(out, out0 + w.float() @ v, 1e-5, 1e-4, "matvec_bf16")
    rows = 37
    zn = rnd((rows + 5, D), g, 0.06).to(BF)
    d_rows = torch.tensor([rows // 2 + 1], dtype=torch.int32)            # rows_mul * d_rows = 38 > 37 live rows is clamped by the caller's buffer
    live = min(2 * int(d_rows[0]), zn.shape[0])
    zsum = ops.colsum_bf16(zn.to(dev), torch.zeros(D, device=dev), d_rows=d_rows.to(dev), rows_mul=2)
    cs = ops.matvec_bf16(w.to(dev), zsum, torch.zeros(K, device=dev))
    logits = zn.float()[:live] @ w.float().t()
    close(cs, logits.sum(0), 1e-4, 1e-4, "column sums of zn @ w^T through the factors")


def check_mlp_bwd_fused(dev, M, E, H, rps=128, seed=41):
    """ccd_mlp_bwd_fused against autograd of the same chain in fp32 (on the bf16-rounded operands): du, the LayerNorm-2 backward on the
    bf16 gradient stream (g, gb_out, dgamma, dbeta, dbias) and the fc1.bias gradient - and against the two launches it replaces
    (gelu'(u) product + LayerNorm-backward product).  vision_transformer.py:59-65,110 in autograd order."""
    g_ = torch.Generator().manual_seed(seed)
    gb = rnd((M, E), g_, 0.5).to(BF)
    w2 = rnd((E, H), g_, 0.06).to(BF)                 # fc2.weight [E, H]
    w1 = rnd((H, E), g_, 0.08).to(BF)                 # fc1.weight [H, E]
    u = rnd((M, H), g_, 1.5).to(BF)
    u[0, :8] = torch.tensor([0.0, -0.0, 1e-3, -1e-3, 20.0, -20.0, 5.0, -5.0]).to(BF)      # table edges
    x = rnd((M, E), g_); gamma = 1.0 + 0.2 * rnd((E,), g_)
    g0 = rnd((M, E), g_).to(BF).float()
    nsamp = (M + rps - 1) // rps
    rowscale = (torch.rand(nsamp, generator=g_) > 0.3).float() * 1.25
    mean = x.mean(1); rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
    # reference chain
    dh = gb.float() @ w2.float()
    uf = u.float().double().requires_grad_(True)
    F.gelu(uf).sum().backward()
    du_ref = (dh * uf.grad.float())
    dy2 = du_ref.to(BF).float() @ w1.float()          # (the kernel feeds the bf16-rounded du to the second product)
    xh = (x - mean[:, None]) * rstd[:, None]
    dg = dy2 * gamma
    dx = (dg - dg.mean(1, keepdim=True) - xh * (dg * xh).mean(1, keepdim=True)) * rstd[:, None]
    g_ref = g0 + dx
    rs_rows = rowscale.repeat_interleave(rps)[:M, None]
    wt2 = w2.t().contiguous().to(dev)                 # fc2.weight^T [H, E]
    wt1 = w1.t().contiguous().to(dev)                 # fc1.weight^T [E, H]
    gdev = g0.to(BF).to(dev)
    dgam, dbet = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    dbias, db1 = torch.zeros(E, device=dev), torch.full((H,), 0.25, device=dev)
    gb_out = torch.empty((M, E), dtype=BF, device=dev)
    du = ops.mlp_bwd_fused(gb.to(dev), wt2, wt1, u.to(dev), db1=db1, x=x.to(dev), mean=mean.to(dev), rstd=rstd.to(dev),
                           gamma=gamma.to(dev), g=gdev, dgamma=dgam, dbeta=dbet, gb_out=gb_out, rowscale=rowscale.to(dev),
                           rows_per_sample=rps, dbias=dbias)
    close(du, du_ref, 2e-2, 2e-2 * du_ref.abs().max().item(), "mlp_bwd/du")
    close(gdev, g_ref, 2e-2, 2e-2 * dx.abs().max().item() + 2e-2, "mlp_bwd/g")
    close(gb_out, g_ref * rs_rows, 2e-2, 2e-2 * g_ref.abs().max().item(), "mlp_bwd/gb_out")
    tol = 4e-3 * M ** 0.5
    close(dgam, (dy2 * xh).sum(0), 2e-2, tol * (dy2 * xh).abs().max().item(), "mlp_bwd/dgamma")
    close(dbet, dy2.sum(0), 2e-2, tol * dy2.abs().max().item(), "mlp_bwd/dbeta")
    close(dbias, gb_out.float().cpu().sum(0), 1e-3, 1e-3 * M ** 0.5 * gb_out.float().abs().max().item(), "mlp_bwd/dbias(proj)")
    close(db1 - 0.25, du_ref.sum(0), 2e-2, tol * du_ref.abs().max().item(), "mlp_bwd/db1")
    if H % (64 * (3 if E == 384 else 2)) != 0:         # (the row-owner LayerNorm-backward product takes K in whole rings of k-blocks)
        return
    # the two launches it replaces
    g2 = g0.to(BF).to(dev)
    dg2, db2, dbi2, db12 = torch.zeros(E, device=dev), torch.zeros(E, device=dev), torch.zeros(E, device=dev), torch.zeros(H, device=dev)
    gact2 = torch.empty((M, H), dtype=BF, device=dev)
    du2 = ops.gemm_nt(gb.to(dev), wt2, epilogue=ops.EPI_DGELU, aux=u.to(dev), out2=gact2, colsum=db12)
    gb2 = torch.empty((M, E), dtype=BF, device=dev)
    ops.gemm_nt_lnbwd(du2, wt1, x.to(dev), mean.to(dev), rstd.to(dev), gamma.to(dev), g2, dg2, db2, accumulate=True, gb=gb2,
                      rowscale=rowscale.to(dev), rows_per_sample=rps, dbias=dbi2)
    close(du, du2.float(), 2e-2, 1e-2 * du_ref.abs().max().item(), "mlp_bwd/du vs two-launch")
    close(gdev, g2.float(), 2e-2, 1e-2 * dx.abs().max().item() + 2e-2, "mlp_bwd/g vs two-launch")


def check_gemm_lnbwd(dev, M, N, K, seed=33, g16=False, tap=False):
    """Data-gradient product with the LayerNorm backward in its epilogue == gemm_nt followed by ln_bwd, and both == autograd.
    g16: the gradient stream g is a bf16 tensor (ccd_gemm_nt_lnbwd_g16) - read as bf16, accumulated in fp32, rounded once.
    tap: a second LayerNorm of the same rows (other gamma / beta) whose output gradient arrives as a bf16 tensor and whose backward
    pass joins the epilogue (ccd_gemm_nt_lnbwd_tap_g16) == autograd of the SUM of the two LayerNorms (vision_transformer.py:245-249)."""
    g = torch.Generator().manual_seed(seed)
    a = rnd((M, K), g).to(BF); b = rnd((N, K), g, 0.2).to(BF)
    x = rnd((M, N), g) * 2 + 0.3
    gamma = 1 + 0.1 * rnd((N,), g)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    g0 = rnd((M, N), g)
    if g16:
        g0 = g0.to(BF).float()
    rps = 16
    rowscale = (torch.rand((M + rps - 1) // rps, generator=g) > 0.3).float() * 1.25
    dy = a.float() @ b.float().t()
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = torch.zeros(N, requires_grad=True)
    F.layer_norm(xr, (N,), gr, br, 1e-6).backward(dy)
    if tap:
        gamma_t = 1 + 0.2 * rnd((N,), g)
        d_tap = (rnd((M, N), g) * float(dy.std())).to(BF)
        gtr = gamma_t.clone().requires_grad_(True); btr = torch.zeros(N, requires_grad=True)
        F.layer_norm(xr, (N,), gtr, btr, 1e-6).backward(d_tap.float())          # accumulates onto xr.grad
    scale_dy = float(dy.pow(2).mean().sqrt())
    for acc in (True, False):
        for tail in (True, False):
            gbuf = (g0.to(BF) if g16 else g0.clone()).to(dev)
            dgam = torch.full((N,), 0.5).to(dev); dbet = torch.full((N,), -0.25).to(dev)
            gb = torch.zeros(M, N, dtype=BF).to(dev) if tail else None
            dbias = torch.full((N,), 0.25).to(dev)
            tap_kw = {}
            if tap:
                dgam_t = torch.full((N,), 0.125).to(dev); dbet_t = torch.full((N,), -0.5).to(dev)
                tap_kw["tap"] = (d_tap.to(dev), gamma_t.to(dev), dgam_t, dbet_t)
            ops.gemm_nt_lnbwd(a.to(dev), b.to(dev), x.to(dev), mean.to(dev), rstd.to(dev), gamma.to(dev), gbuf, dgam, dbet,
                              accumulate=acc, gb=gb, rowscale=rowscale.to(dev) if tail else None, rows_per_sample=rps,
                              dbias=dbias if tail else None, **tap_kw)
            want_g = xr.grad + (g0 if acc else 0)
            tag = f"lnbwd acc={acc} tail={tail}"
            close(gbuf, want_g, 1e-2 if g16 else 2e-3, 2e-3 * scale_dy, tag + "/g")
            close(dgam, gr.grad + 0.5, 2e-3, 2e-3 * scale_dy * M ** 0.5, tag + "/dgamma")
            close(dbet, br.grad - 0.25, 2e-3, 2e-3 * scale_dy * M ** 0.5, tag + "/dbeta")
            if tap:
                close(dgam_t, gtr.grad + 0.125, 2e-3, 2e-3 * scale_dy * M ** 0.5, tag + "/tap dgamma")
                close(dbet_t, btr.grad - 0.5, 2e-3, 2e-3 * scale_dy * M ** 0.5, tag + "/tap dbeta")
            if tail:
                want_gb = (want_g * rowscale.repeat_interleave(rps)[:M, None])
                close(gb, want_gb, 1e-2, 1e-2 * scale_dy, tag + "/gb")
                close(dbias, want_gb.to(BF).float().sum(0) + 0.25, 5e-3, 2e-2 * scale_dy * M ** 0.5, tag + "/dbias")


# ------------------------------------------------------------------------------------------------ finetune path
def drop_keep_ref(seed, n, p):
    """Python mirror of decoder.h: drop_keep (splitmix64 finaliser on seed + index)."""
    import numpy as np
    if p <= 0:
        return torch.ones(n, dtype=torch.bool)
    thr = min(int(p * 4294967296.0), 4294967295)
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return torch.from_numpy(((z >> np.uint64(32)) >= np.uint64(thr)))


def check_dropout(dev, seed=40):
    g = torch.Generator().manual_seed(seed)
    n, p, sd = 4096 * 6, 0.1, 0x1234_5678_9abc_def1
    x = rnd((n,), g)
    keep = drop_keep_ref(sd, n, p)
    assert 0.88 < keep.float().mean().item() < 0.92
    want = torch.where(keep, x / (1 - p), torch.zeros(()))
    close(ops.dropout(x.to(dev), p, sd), want, 1e-6, 1e-6, "dropout/f32")
    r = rnd((n,), g)
    close(ops.dropout(x.to(dev), p, sd, resid=r.to(dev)), want + r, 1e-6, 1e-6, "dropout/resid")
    close(ops.dropout(x.to(dev), p, sd, out_dtype=BF), want.to(BF), 1e-6, 1e-6, "dropout/f32->bf16")
    xb = x.to(BF).to(dev)
    ops.dropout(xb, p, sd, out=xb)                                    # in place
    close(xb, torch.where(keep, x.to(BF).float() / (1 - p), torch.zeros(())).to(BF), 1e-6, 1e-6, "dropout/bf16")
    close(ops.dropout(x.to(dev), 0.0, sd), x, 0, 0, "dropout/p0")
    assert (drop_keep_ref(sd + 1, n, p) != keep).any()


def check_droppath(dev):
    keep = torch.tensor([1.0, 0.9, 0.5])
    out = ops.droppath_scales(keep.to(dev), 4000, 99).cpu()
    assert out.shape == (3, 2, 4000) and (out[0] == 1).all()
    for i, k in ((1, 0.9), (2, 0.5)):
        vals = set(out[i].unique().tolist())
        assert vals == {0.0, float(torch.tensor(1.0) / torch.tensor(k))}, vals
        assert abs((out[i] > 0).float().mean().item() - k) < 0.03
    ref = drop_keep_ref(99, 3 * 8000, 0.5).view(3, 2, 4000)
    assert torch.equal(out[2] > 0, ref[2])
    assert not torch.equal(out, ops.droppath_scales(keep.to(dev), 4000, 100).cpu())
    # device-side seed offset (graph replays): seed 90 at launch + 9 on the device == seed 99; wraps modulo 2^64
    assert torch.equal(out, ops.droppath_scales(keep.to(dev), 4000, 90, torch.tensor([9], dtype=torch.int64).to(dev)).cpu())
    assert torch.equal(out, ops.droppath_scales(keep.to(dev), 4000, 100, torch.tensor([-1], dtype=torch.int64).to(dev)).cpu())


def check_dec_embed(dev, B=5, T=25, D=128, C=93, seed=41):
    g = torch.Generator().manual_seed(seed)
    tok = torch.randint(0, C - 1, (B, T), generator=g)
    tok[:, 0] = 91
    tok[:, T // 2:] = 92                                              # padding rows
    emb, pos = rnd((C, D), g), rnd((40, D), g)
    want = emb[tok] + pos[:T]
    close(ops.dec_embed_fwd(tok.to(dev), emb.to(dev), pos.to(dev)).view(B, T, D), want, 1e-6, 1e-6, "embed/fwd")
    p, sd = 0.25, 77
    keep = drop_keep_ref(sd, B * T * D, p).view(B, T, D)
    close(ops.dec_embed_fwd(tok.to(dev), emb.to(dev), pos.to(dev), p, sd).view(B, T, D),
          torch.where(keep, want / (1 - p), torch.zeros(())), 1e-6, 1e-6, "embed/fwd-drop")
    dx = rnd((B * T, D), g)
    demb = torch.full((C, D), 0.5).to(dev)
    ops.dec_embed_bwd(tok.to(dev), dx.to(dev), demb, 92, p, sd)
    ref = torch.zeros(C, D)
    ref.index_add_(0, tok.view(-1), torch.where(keep.view(B * T, D), dx / (1 - p), torch.zeros(())))
    ref[92] = 0
    close(demb, ref + 0.5, 1e-5, 1e-5, "embed/bwd")


def _dec_attn_ref(q, k, v, mask, scale, keep=None, keep_scale=1.0):
    """q [B,H,Tq,64], k/v [B,H,Tk,64] fp32; mask [B,1,Tq,Tk] bool or None -> (out, probs after dropout)."""
    s = (q * scale) @ k.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = pr * keep * keep_scale
    return pr @ v, pr


def check_dec_attn(dev, B=3, H=2, Tq=25, Tk=25, self_attn=True, p=0.0, seed=42):
    g = torch.Generator().manual_seed(seed)
    E = 64 * H
    ld = 3 * E + 8                                                    # exercises the row strides
    qkv = rnd((B * max(Tq, Tk), ld), g).to(BF)
    if self_attn:
        qv, kv_, vv = qkv[:B * Tq, 0:E], qkv[:B * Tk, E:2 * E], qkv[:B * Tk, 2 * E:3 * E]
    else:
        qv = rnd((B * Tq, E), g).to(BF)
        kvb = rnd((B * Tk, 2 * E + 16), g).to(BF)
        kv_, vv = kvb[:, 0:E], kvb[:, E:2 * E]
    tok = mask = None
    if self_attn:
        tok = torch.randint(0, 90, (B, Tk), generator=g)
        for b in range(B):
            tok[b, 3 + 5 * b:] = 92
        pad = (tok != 92)[:, None, None, :]
        causal = torch.tril(torch.ones(Tq, Tk)).bool()[None, None]
        mask = pad & causal
    sd = 991
    keep = drop_keep_ref(sd, B * H * Tq * Tk, p).view(B, H, Tq, Tk).float() if p > 0 else None
    sp = lambda t_, T: t_.float().reshape(B, T, H, 64).transpose(1, 2).clone().requires_grad_(True)
    qf, kf, vf = sp(qv, Tq), sp(kv_, Tk), sp(vv, Tk)
    out_ref, pr_ref = _dec_attn_ref(qf, kf, vf, mask, 0.125, keep, 1.0 / (1.0 - p))
    kw = dict(tokens=None if tok is None else tok.to(dev), pad_idx=92, causal=self_attn, p=p, seed=sd)
    qd, kd, vd = (qkv.to(dev)[:B * Tq, 0:E], qkv.to(dev)[:B * Tk, E:2 * E], qkv.to(dev)[:B * Tk, 2 * E:3 * E]) if self_attn \
        else (qv.to(dev), kvb.to(dev)[:, 0:E], kvb.to(dev)[:, E:2 * E])
    out, lse, probs = ops.dec_attn_fwd(qd, kd, vd, B, H, Tq, Tk, 0.125, want_probs=True, **kw)
    close(probs, pr_ref, 2e-3, 1e-5, "dec_attn/probs")
    close(out.view(B, Tq, H, 64).transpose(1, 2), out_ref, 1e-2, 1e-2, "dec_attn/out")
    d_out = rnd((B * Tq, E), g).to(BF)
    out_ref.backward(d_out.float().view(B, Tq, H, 64).transpose(1, 2))
    dq = torch.zeros((B * Tq, E + 8), dtype=BF).to(dev)
    dkv = torch.zeros((B * Tk, 2 * E + 24), dtype=BF).to(dev)
    ops.dec_attn_bwd(qd, kd, vd, out, d_out.to(dev), lse, dq[:, :E], dkv[:, :E], dkv[:, E:2 * E], B, H, Tq, Tk, 0.125, **kw)
    un = lambda t_, T: t_.transpose(1, 2).reshape(B * T, E)
    tol = lambda w: 2e-2 * float(w.abs().max()) + 1e-6
    close(dq[:, :E], un(qf.grad, Tq), 2e-2, tol(qf.grad), "dec_attn/dq")
    close(dkv[:, :E], un(kf.grad, Tk), 2e-2, tol(kf.grad), "dec_attn/dk")
    close(dkv[:, E:2 * E], un(vf.grad, Tk), 2e-2, tol(vf.grad), "dec_attn/dv")
    assert (dq[:, E:] == 0).all() and (dkv[:, 2 * E:] == 0).all()


def check_tf_loss(dev, B=6, T=25, C=92, seed=43):
    g = torch.Generator().manual_seed(seed)
    ld = 128
    logits = rnd((B * T, ld), g, 2.0)
    tgt = torch.randint(0, C, (B, T), generator=g)
    tgt[:, 0] = 91
    for b in range(B):
        tgt[b, 4 + 3 * b:] = 92
    lg = logits[:, :C].clone().view(B, T, C).requires_grad_(True)
    ref = F.cross_entropy(lg[:, :-1].reshape(-1, C), tgt[:, 1:].reshape(-1), ignore_index=92)
    ref.backward()
    row_lse, acc = ops.tf_loss_fwd(logits.to(dev), C, tgt.to(dev), 92)
    close(acc[0] / acc[1], ref, 1e-5, 1e-6, "tf_loss/fwd")
    assert float(acc[1]) == float((tgt[:, 1:] != 92).sum())
    d = ops.tf_loss_bwd(logits.to(dev), C, tgt.to(dev), 92, row_lse, acc, torch.tensor([0.5]).to(dev), 128)
    close(d[:, :C].view(B, T, C), 0.5 * lg.grad, 1e-2, 1e-5, "tf_loss/bwd")
    assert (d[:, C:] == 0).all()
    # greedy step
    probs = torch.zeros(B, T, C).to(dev)
    seq = torch.full((B, T + 1), 92, dtype=torch.int64).to(dev)
    lg2 = logits[:B].clone()
    lg2[1, 7] = lg2[1, 70] = 50.0                                      # tie: first index wins
    ops.greedy_step(lg2.to(dev), C, probs, 3, seq)
    close(probs[:, 3], torch.softmax(lg2[:, :C], -1), 1e-5, 1e-7, "greedy/probs")
    assert torch.equal(seq[:, 4].cpu(), lg2[:, :C].argmax(-1)) and int(seq[1, 4]) == 7
    assert (seq[:, :4] == 92).all() and (seq[:, 5:] == 92).all()


def check_decoder_pieces(dev):
    with ops.policy(dec_attn_simt=1):                                  # the general (masked / short) kernels on the 256-key case
        check_dec_attn(dev, B=1, H=2, Tq=25, Tk=256, self_attn=False, p=0.1)
        check_dec_attn(dev, B=2, H=2, Tq=26, Tk=26, self_attn=True, p=0.1)
    check_dropout(dev)
    check_droppath(dev)
    check_dec_embed(dev)
    check_dec_embed(dev, B=50, D=64)                                   # 1250 rows: two row chunks per class
    check_dec_attn(dev, self_attn=True)
    check_dec_attn(dev, B=2, H=2, Tq=26, Tk=26, self_attn=True, p=0.1)
    check_dec_attn(dev, B=3, H=3, Tq=32, Tk=32, self_attn=True, p=0.1)  # 9 (sample, head) items: a ragged last workgroup
    check_dec_attn(dev, B=2, H=3, Tq=25, Tk=256, self_attn=False)
    check_dec_attn(dev, B=1, H=2, Tq=25, Tk=256, self_attn=False, p=0.1)
    check_tf_loss(dev)
