"""Parity of every HIP kernel on a real MI355X, through the C ABI of libccd_hip.so (run with -m gpu)."""
import pytest
import torch

from backends import Backend
import kernel_checks as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    with Backend("hip") as b:
        yield b


@pytest.mark.parametrize("M,N,K", [(200, 136, 128), (1024, 1152, 384), (4096, 384, 1536), (96, 65536, 256),
                                   (1000, 192, 192), (40000, 264, 192), (33000, 1536, 384), (131072, 384, 384), (33000, 1152, 384)])
def test_gemm_nt(hip, M, N, K):
    kc.check_gemm_nt(hip.device, M, N, K)


@pytest.mark.parametrize("Mc,P,Q,splits", [(300, 264, 72, 3), (131072, 1152, 384, 0), (4096, 384, 384, 0)])
def test_gemm_tn_colsum(hip, Mc, P, Q, splits):
    kc.check_gemm_tn_colsum(hip.device, Mc, P, Q, splits)


def test_gemm_nt_split_k(hip):
    kc.check_gemm_nt_split_k(hip.device)
    kc.check_gemm_nt_split_k(hip.device, M=3000, N=256, K=65536)


@pytest.mark.parametrize("Mc,P,Q,splits", [(300, 136, 72, 3), (4096, 1152, 384, 0), (8192, 384, 1536, 0),
                                           (96, 65536, 256, 1), (64, 8, 264, 0)])
def test_gemm_tn(hip, Mc, P, Q, splits):
    kc.check_gemm_tn(hip.device, Mc, P, Q, splits=splits)


@pytest.mark.parametrize("Mc,shape1,shape2", [(8192, (384, 1536), (1536, 384)), (4096 + 64, (384, 384), (1152, 384)),
                                              (8192 + 32, (512, 2048), (2048, 512)), (65536, (512, 512), (1536, 512)),
                                              (131072, (384, 384), (1152, 384)), (600, (136, 72), (384, 192))])
def test_gemm_tn_pair(hip, Mc, shape1, shape2):
    """ccd_gemm_tn_pair: the MLP pair and the attention pair in gemm_tn384.h's grouped launch (16 / 8 tiles per contraction slice,
    ragged slices, the full 131072 rows of the benchmark batch), the E = 512 pairs (two-call fallback by default, 512 x 128 tiles
    with gemm_tn384_geom = 2), and the two-call fallback for other shapes."""
    from ccd_amd import ops
    kc.check_gemm_tn_pair(hip.device, Mc, shape1, shape2)
    with ops.policy(gemm_tn384_geom=2):
        kc.check_gemm_tn_pair(hip.device, Mc, shape1, shape2, seed=15)


def test_gemm_tn_pair_reserved_cus(hip):
    """The grouped launch with compute units held back for the collectives (the N > 1 engine sets cu_reserve = 8: 31 slots per
    XCD): the two problems' groups are placed independently - 2 + 1 / 1 + 2 groups of 8 for the MLP pair, 3-5 + 3-4 for the
    attention pair - with different slice counts per problem; an odd reserve leaves XCDs with different slot counts."""
    from ccd_amd import ops
    for reserve in (8, 13):
        with ops.policy(cu_reserve=reserve):
            kc.check_gemm_tn_pair(hip.device, 16384 + 96, (384, 1536), (1536, 384), seed=11)
            kc.check_gemm_tn_pair(hip.device, 16384, (384, 384), (1152, 384), seed=12)
            kc.check_gemm_tn(hip.device, 8192, 1536, 384, seed=13)


@pytest.mark.parametrize("policy", [dict(gemm_tn384=0), dict(gemm_tn384_min_tiles=1)])
def test_gemm_tn_policies(hip, policy):
    """Single weight-gradient products on the kernel that is not the default for their shape: the 128-square kernel for the
    fc shapes, gemm_tn384.h for proj (2 tiles, 128 slices)."""
    from ccd_amd import ops
    with ops.policy(**policy):
        kc.check_gemm_tn(hip.device, 8192, 384, 384)
        kc.check_gemm_tn(hip.device, 4096, 1536, 384, seed=7)
        kc.check_gemm_tn_pair(hip.device, 8192 + 32, (384, 1536), (1536, 384), seed=8)


@pytest.mark.parametrize("M,N,K", [(300, 128, 384), (40000, 1152, 384), (131072, 384, 384), (70000 + 17, 1536, 512),
                                   (16384, 64, 384)])
def test_rowproj(hip, M, N, K):
    """rowproj.h: the K = 384 / 512 bf16 projections with the activation rows resident in registers (qkv, proj data gradient),
    ragged last tiles, strided operands, several tiles per workgroup; against fp32 torch and against the tiled kernels."""
    kc.check_rowproj(hip.device, M=M, N=N, K=K, seed=M % 7)


def test_rowproj_half_height_tiles(hip):
    """K = 384 with one row block per wave: chosen when 256-row tiles would leave compute units idle (32 768 rows = 64 images), forced here."""
    from ccd_amd import ops
    kc.check_rowproj(hip.device, M=32768, N=1152, K=384, seed=2)          # auto: 128 tiles of 256 rows < 256 CUs -> 128-row tiles
    with ops.policy(rowproj_rb=1):
        kc.check_rowproj(hip.device, M=40000, N=384, K=384, seed=3)
    with ops.policy(rowproj_rb=2):
        kc.check_rowproj(hip.device, M=32768, N=1152, K=384, seed=2)


def test_rowproj_is_the_default_for_the_vit_projections(hip):
    """At the benchmark shapes ccd_gemm_nt (EPI_BF16, K = 384, 131072 rows) runs rowproj.h: 20 launches must be bit-identical
    (a row-owner kernel has no split-K atomics) and equal the tiled kernel to bf16 rounding."""
    import torch
    from ccd_amd import ops
    g = torch.Generator().manual_seed(3)
    a = kc.rnd((131072, 384), g).to(kc.BF).to(hip.device)
    w = kc.rnd((1152, 384), g, 0.2).to(kc.BF).to(hip.device)
    bias = kc.rnd((1152,), g).to(hip.device)
    assert ops.policy_get("rowproj") == 1 and ops.policy_get("rowproj_min_m") <= 131072
    first = ops.gemm_nt(a, w, bias=bias)
    for _ in range(20):
        assert torch.equal(ops.gemm_nt(a, w, bias=bias), first)
    with ops.policy(rowproj=0):
        tiled = ops.gemm_nt(a, w, bias=bias)
    kc.close(first, tiled.float(), 1e-2, 1e-2, "rowproj vs gemm256 at the qkv shape")


@pytest.mark.parametrize("rows,E", [(37, 192), (4096, 384), (1001, 512)])
def test_layernorm(hip, rows, E):
    kc.check_layernorm(hip.device, rows, E)
    kc.check_layernorm(hip.device, rows, E, g16=True)


@pytest.mark.parametrize("views,heads,spike", [(1, 2, False), (16, 6, False), (3, 8, True)])
def test_attention(hip, views, heads, spike):
    from ccd_amd import ops
    kc.check_attention(hip.device, views, heads, spike=spike)
    with ops.policy(attn_tr=0):                               # dK / dV on the four register-staged images
        kc.check_attention(hip.device, views, heads, spike=spike)
    with ops.policy(cu_reserve=248):                          # 8 workgroups walk the blocks: the double-buffered images turn over
        kc.check_attention(hip.device, views, heads, spike=spike)


def test_gemm_dynamic_rows(hip):
    from ccd_amd import ops
    g = torch.Generator().manual_seed(0)
    a = kc.rnd((3000, 256), g).to(kc.BF); b = kc.rnd((520, 256), g).to(kc.BF)
    d_rows = torch.tensor([750], dtype=torch.int32, device=hip.device)
    out = torch.full((3000, 520), 7.0, device=hip.device)
    ops.gemm_nt(a.to(hip.device), b.to(hip.device), epilogue=ops.EPI_F32, out=out, d_rows=d_rows, rows_mul=2)
    kc.close(out[:1500], a[:1500].float() @ b.float().t(), 1e-4, 1e-3, "dyn/nt")
    assert (out[1500:] == 7.0).all()
    c = kc.rnd((3000, 136), g).to(kc.BF)
    acc = torch.zeros(256, 136, device=hip.device)
    ops.gemm_tn(a.to(hip.device), c.to(hip.device), acc, d_rows=d_rows, rows_mul=2)
    kc.close(acc, a[:1500].float().t() @ c[:1500].float(), 1e-4, 5e-2, "dyn/tn")


def test_ccl(hip):
    kc.check_ccl(hip.device)



def test_seg_to_mask(hip):
    kc.check_seg_to_mask(hip.device)
    kc.check_predicted_mask_chain(hip.device)


def test_warp(hip):
    kc.check_warp(hip.device)


@pytest.mark.parametrize("E", [128, 384])
def test_region(hip, E):
    kc.check_region(hip.device, E=E)
    kc.check_region_adjacent_planes(hip.device)


@pytest.mark.parametrize("views,E", [(3, 192), (32, 384)])
def test_patch_embed(hip, views, E):
    kc.check_patch_embed(hip.device, views=views, E=E)


def test_small_ops(hip):
    kc.check_small_ops(hip.device)


@pytest.mark.parametrize("rows,D,K", [(37, 64, 200), (500, 256, 4096)])
def test_head_pieces(hip, rows, D, K):
    kc.check_head_pieces(hip.device, rows=rows, D=D, K=K)


@pytest.mark.parametrize("M,K", [(11, 4096), (48, 65536)])
def test_dino_loss(hip, M, K):
    kc.check_dino_loss(hip.device, M=M, K=K)


@pytest.mark.parametrize("M,K", [(11, 4096), (48, 65536), (1650, 65536), (300, 1536)])
def test_head_loss(hip, M, K):
    """ccd_head_loss_fwd / _bwd: last layer + distillation loss with the logits in registers (Dino_loss.py:81-105 on
    vision_transformer.py:326-327's product) against the oracle and against the unfused kernels; (1650, 65536) is the benchmark's size."""
    kc.check_head_loss(hip.device, M=M, K=K, seed=M % 13)


def test_seg_loss(hip):
    kc.check_seg_loss(hip.device, half=4)


def test_optimizer(hip):
    kc.check_optimizer(hip.device)


def test_conv_pieces(hip):
    kc.check_conv_pieces(hip.device)


def test_decoder_pieces(hip):
    """finetune path: dropout, target embedding, short-query attention (masks, dropout), TFLoss, greedy step."""
    kc.check_decoder_pieces(hip.device)
    kc.check_dec_attn(hip.device, B=64, H=8, Tq=25, Tk=256, self_attn=False, p=0.1)
    kc.check_dec_attn(hip.device, B=64, H=8, Tq=25, Tk=25, self_attn=True, p=0.1)


@pytest.mark.parametrize("images,E", [(2, 192), (8, 384)])
def test_seghead(hip, images, E):
    kc.check_seghead(hip.device, images=images, E=E)


@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (4096, 1152, 384), (2048, 384, 1536)])
def test_gemm256(hip, M, N, K):
    """256x256 LDS-DMA kernel: forced onto a ragged small problem and on model shapes (repeated: the DMA is async)."""
    from ccd_amd import ops
    with ops.policy(gemm_256_min_m=1, gemm_256_min_n=1):
        for seed in range(3):
            kc.check_gemm_nt(hip.device, M=M, N=N, K=K, seed=seed)
        kc.check_gemm_dynamic_rows(hip.device, M=max(M, 600), N=N, K=K, live=75)
        with ops.policy(gemm_256_deep=1):                         # BK = 32 x 4 buffers, counted vmcnt (async: repeat)
            for seed in range(4):
                kc.check_gemm_nt(hip.device, M=M, N=N, K=K, seed=seed)
            kc.check_gemm_dynamic_rows(hip.device, M=max(M, 600), N=N, K=K, live=75)
    with ops.policy(gemm_256=2, gemm_256_min_m=1, gemm_256_min_n=1000000):   # the 256x128 variant, every epilogue
        for seed in range(2):
            kc.check_gemm_nt(hip.device, M=M, N=N, K=K, seed=seed)
        kc.check_gemm_dynamic_rows(hip.device, M=max(M, 600), N=N, K=K, live=75)


@pytest.mark.parametrize("M,N,K", [(300, 136, 128), (4096, 384, 1536), (2048, 384, 384)])
def test_gemm_row384(hip, M, N, K):
    from ccd_amd import ops
    with ops.policy(gemm_256_min_m=1, gemm_row384=2):
        for seed in range(2):
            kc.check_gemm_nt(hip.device, M=M, N=N, K=K, seed=seed)
        kc.check_gemm_dynamic_rows(hip.device, M=max(M, 600), N=N, K=K, live=75)


@pytest.mark.parametrize("M,N,K", [(300, 384, 128), (300, 384, 384), (4096, 384, 1536), (40000, 384, 1152), (2048, 192, 768),
                                   (5000, 128, 512), (33000, 256, 768), (20000, 512, 1536), (4100, 512, 2048)])
def test_gemm_lnbwd(hip, M, N, K):
    from ccd_amd import ops
    kc.check_gemm_lnbwd(hip.device, M=M, N=N, K=K)            # rowgemm.h where K % (64 R) == 0 and N in {128, 256, 384, 512}


@pytest.mark.parametrize("M,N,K", [(300, 384, 384), (4096, 384, 1536), (40000, 384, 1152), (5000, 128, 512), (33000, 256, 768)])
def test_gemm_lnbwd_bf16_stream(hip, M, N, K):
    """ccd_gemm_nt_lnbwd_g16: the residual-gradient stream as a bf16 tensor (read bf16, accumulated fp32, rounded once per writer)."""
    from ccd_amd import ops
    kc.check_gemm_lnbwd(hip.device, M=M, N=N, K=K, g16=True)
    if N == 384:
        with ops.policy(rowgemm_adma=0):
            kc.check_gemm_lnbwd(hip.device, M=M, N=N, K=K, seed=36, g16=True)


@pytest.mark.parametrize("M,K", [(300, 384), (4096, 1536), (40000, 1152), (131072 + 40, 1152)])
def test_gemm_lnbwd_with_tap(hip, M, K):
    """ccd_gemm_nt_lnbwd_tap_g16: a segmentation tap's LayerNorm backward inside the epilogue of the qkv data-gradient product."""
    from ccd_amd import ops
    kc.check_gemm_lnbwd(hip.device, M=M, N=384, K=K, seed=41, g16=True, tap=True)
    if M <= 4096:
        with ops.policy(rowgemm_adma=0):
            kc.check_gemm_lnbwd(hip.device, M=M, N=384, K=K, seed=42, g16=True, tap=True)


@pytest.mark.parametrize("M,E,H,rps", [(300, 384, 128, 8), (4096 + 72, 384, 1536, 256), (131072, 384, 1536, 256), (33000, 256, 1024, 256)])
def test_mlp_bwd_fused(hip, M, E, H, rps):
    """The MLP branch's data-gradient chain in one launch (mlp_bwd.h): gelu'(u) product, fc1 data gradient, LayerNorm-2 backward, du for
    the weight gradients, fc1.bias column sums - ragged tiles, many tiles per workgroup, the benchmark shape."""
    kc.check_mlp_bwd_fused(hip.device, M=M, E=E, H=H, rps=rps, seed=M % 11)


def test_mlp_bwd_fused_repeatable(hip):
    """Counted vmcnt + barriers order the weight ring; the u rows arrive through the same queue: 10 launches must agree bit for bit in
    everything that is not an atomic sum (du, g, gb_out)."""
    import torch
    from ccd_amd import ops
    g = torch.Generator().manual_seed(9)
    M, E, H = 131072, 384, 1536
    dev = hip.device
    gb = kc.rnd((M, E), g, 0.5).to(kc.BF).to(dev); u = kc.rnd((M, H), g, 1.5).to(kc.BF).to(dev)
    w2t = kc.rnd((H, E), g, 0.06).to(kc.BF).to(dev); w1t = kc.rnd((E, H), g, 0.08).to(kc.BF).to(dev)
    x = kc.rnd((M, E), g).to(dev); gamma = torch.ones(E, device=dev)
    mean = x.mean(1); rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
    g0 = kc.rnd((M, E), g).to(kc.BF).to(dev)
    ref = None
    for _ in range(10):
        gg = g0.clone()
        gbo = torch.empty((M, E), dtype=kc.BF, device=dev)
        du = ops.mlp_bwd_fused(gb, w2t, w1t, u, db1=torch.zeros(H, device=dev), x=x, mean=mean, rstd=rstd, gamma=gamma, g=gg,
                               dgamma=torch.zeros(E, device=dev), dbeta=torch.zeros(E, device=dev), gb_out=gbo, rowscale=None,
                               rows_per_sample=256, dbias=torch.zeros(E, device=dev))
        cur = (du.clone(), gg, gbo)
        if ref is None:
            ref = cur
        else:
            for a, b in zip(ref, cur):
                assert torch.equal(a, b)


def test_gemm_lnbwd_row384_kernel(hip):
    from ccd_amd import ops
    with ops.policy(rowgemm=0):
        kc.check_gemm_lnbwd(hip.device, M=4096, N=384, K=1536)


@pytest.mark.parametrize("M,N,K", [(300, 384, 128), (4096, 384, 1536), (2048, 192, 768), (40000, 384, 384), (5000, 128, 512),
                                   (33000, 256, 768), (20000, 512, 512), (4100, 512, 2048)])
def test_gemm_resid_ln(hip, M, N, K):
    from ccd_amd import ops
    kc.check_gemm_resid_ln(hip.device, M=M, N=N, K=K)          # gemm_row384.h for N <= 384, rowgemm.h at N = 512
    if N in (128, 256, 384):
        with ops.policy(rowgemm=2):                            # the row-owner kernel forced
            kc.check_gemm_resid_ln(hip.device, M=M, N=N, K=K)


def test_gemm_resid_ln_policy_off_rejects_n512(hip):
    from ccd_amd import ops
    with ops.policy(rowgemm=0):
        kc.check_gemm_resid_ln(hip.device, M=4096, N=384, K=1536)
        with pytest.raises(RuntimeError):
            kc.check_gemm_resid_ln(hip.device, M=256, N=512, K=512)


@pytest.mark.parametrize("M,E,H,rps", [(128 * 5 + 40, 128, 256, 128), (300, 384, 128, 128), (40000, 384, 1536, 256), (4096, 256, 1024, 256)])
def test_proj_mlp_fused(hip, M, E, H, rps):
    """proj + residual + LayerNorm-2 + fc1 + GELU + fc2 + residual + LayerNorm in one launch (ccd_proj_mlp_fused) vs fp32 torch and
    vs the two launches it replaces: ragged tiles, many tiles per workgroup, every combination of dropped branches (the ring
    skips their weight pieces), with and without the tensors saved for the backward pass."""
    for save in (True, False):
        kc.check_proj_mlp_fused(hip.device, M=M, E=E, H=H, rps=rps, save=save)
    with pytest.raises(RuntimeError):            # a DropPath scale with tiles that span samples: the caller takes the two launches
        kc.check_proj_mlp_fused(hip.device, M=256, E=384, H=128, rps=8)


def test_matvec_bf16(hip):
    kc.check_matvec_bf16(hip.device)
    kc.check_matvec_bf16(hip.device, K=65536, D=256, seed=53)


def test_proj_mlp_fused_repeatable(hip):
    import torch
    from ccd_amd import ops
    g = torch.Generator().manual_seed(5)
    dev = hip.device
    M, E, H = 16384, 384, 1536
    mk = lambda *s, dt=torch.bfloat16, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dt).to(dev)
    a, wp, w1, w2 = mk(M, E), mk(E, E, sc=0.05), mk(H, E, sc=0.05), mk(E, H, sc=0.05)
    bp, b1, b2, ga2, be2, ga, be = (mk(n, dt=torch.float32) for n in (E, H, E, E, E, E, E))
    resid = mk(M, E, dt=torch.float32)
    ref = None
    for _ in range(4):
        out, yn, mean, rstd, kept = ops.proj_mlp_fused(a, wp, bp, resid=resid, rowscale1=None, gamma2=ga2, beta2=be2, w1=w1, b1=b1, w2=w2,
                                                       b2=b2, rowscale2=None, rows_per_sample=256, gamma=ga, beta=be, eps=1e-6, save=True)
        cur = (out, yn, mean, rstd) + kept
        if ref is None:
            ref = [t.clone() for t in cur]
        else:
            assert all(torch.equal(x, y) for x, y in zip(ref, cur)), "proj_mlp_fused is not run-to-run deterministic"


@pytest.mark.parametrize("M,E,H,rps", [(300, 128, 256, 128), (200, 384, 128, 8), (40000, 384, 1536, 256), (4096, 256, 1024, 256),
                                       (33000, 512, 2048, 256)])
def test_mlp_fused(hip, M, E, H, rps):
    """fc1 + GELU + fc2 + residual + LayerNorm in one launch: ragged tiles, many tiles per workgroup (the ring of weight
    pieces runs across them), dropped samples, with and without the stored pre-activation."""
    for store_u in (True, False):
        kc.check_mlp_fused(hip.device, M=M, E=E, H=H, rps=rps, store_u=store_u)


def test_mlp_fused_repeatable(hip):
    """The weight ring is ordered by counted vmcnt + barriers only: 20 launches on the same operands must agree bit for bit
    (a piece read before its DMA landed would show up as run-to-run differences)."""
    from ccd_amd import ops
    g = torch.Generator().manual_seed(5)
    M, E, H = 131072, 384, 1536
    dev = hip.device
    y = kc.rnd((M, E), g).to(kc.BF).to(dev); w1 = kc.rnd((H, E), g, 0.08).to(kc.BF).to(dev)
    w2 = kc.rnd((E, H), g, 0.05).to(kc.BF).to(dev)
    b1, b2 = kc.rnd((H,), g).to(dev), kc.rnd((E,), g).to(dev)
    resid = kc.rnd((M, E), g).to(dev); gamma, beta = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    ref = None
    for _ in range(20):
        out, yn, mean, rstd, u = ops.mlp_fused(y, w1, b1, w2, b2, resid=resid, rowscale=None, rows_per_sample=256,
                                               gamma=gamma, beta=beta, eps=1e-6, store_u=True)
        cur = (out.clone(), yn.clone(), u.clone())
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, cur)), "mlp_fused is not run-to-run deterministic"


def test_kmeans2_mask(hip):
    kc.check_kmeans2_mask(hip.device)


def test_multi_launch_helpers(hip):
    kc.check_multi_launch_helpers(hip.device)


def test_cls_tail(hip):
    kc.check_cls_tail(hip.device, images=2)
    kc.check_cls_tail(hip.device, images=67, seed=62, ld_pad=8)             # more bands than the grid takes at once; padded pitch


def test_augment_views(hip):
    kc.check_augment_views(hip.device)
