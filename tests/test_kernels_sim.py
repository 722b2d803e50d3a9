"""Kernel logic under the CPU SIMT executor (tests/hipsim): the product's kernel sources + C ABI compiled for the
host, tiny shapes.  Validates indexing / layouts / epilogues here, where there is no GPU."""
import os

import pytest

from backends import Backend
import kernel_checks as kc


@pytest.fixture(scope="module")
def sim():
    with Backend("sim") as b:
        yield b


@pytest.fixture(autouse=True)
def one_cu(monkeypatch):
    """Kernel tests run on a 1-CU chip: persistent kernels then walk several work items per workgroup."""
    monkeypatch.setenv("CCD_SIM_CUS", "1")


def test_gemm_nt_sim(sim):
    kc.check_gemm_nt(sim.device, M=200, N=136, K=128)


def test_gemm_nt_split_k_sim(sim):
    kc.check_gemm_nt_split_k(sim.device, M=70, N=40)


def test_gemm_tn_colsum_sim(sim):
    kc.check_gemm_tn_colsum(sim.device)
    kc.check_gemm_tn_colsum(sim.device, Mc=64, P=8, Q=264, splits=0)


def test_gemm_tn_sim(sim):
    kc.check_gemm_tn(sim.device, Mc=300, P=136, Q=72, splits=3)
    kc.check_gemm_tn(sim.device, Mc=64, P=8, Q=264)


def test_gemm_tn_pair_sim(sim, monkeypatch):
    """ccd_gemm_tn_pair: both products in gemm_tn384.h's launch (3 tiles per group on a 4-CU chip: 1 + 2), and the fallback to
    two separate products for shapes that kernel does not take."""
    monkeypatch.setenv("CCD_SIM_CUS", "4")
    kc.check_gemm_tn_pair(sim.device, 2048, (384, 192), (384, 384))
    kc.check_gemm_tn_pair(sim.device, 320, (136, 72), (8, 264), seed=6)
    monkeypatch.setenv("CCD_SIM_CUS", "16")    # 8 "XCDs" x 2 slots: the XCDs hold different numbers of groups of either problem
    kc.check_gemm_tn_pair(sim.device, 2048 + 32, (384, 192), (384, 384), seed=16)


def test_gemm_tn512_sim(sim, monkeypatch):
    """gemm_tn384.h with 512 x 128 tiles (4 x 2 MFMA tiles per wave, 3 LDS buffers): the shapes 384 x 192 tiles do not divide
    (vit_base, E = 512) - one tile; a pair of 1 + 2 tiles; 2 x 2 tiles with ragged slices."""
    from ccd_amd import ops
    with ops.policy(gemm_tn384_geom=2, gemm_tn384_min_tiles=1):
        monkeypatch.setenv("CCD_SIM_CUS", "4")
        kc.check_gemm_tn(sim.device, Mc=2048 + 64, P=512, Q=128, seed=21)
        kc.check_gemm_tn_pair(sim.device, 2048, (512, 128), (512, 256), seed=22)
        monkeypatch.setenv("CCD_SIM_CUS", "8")
        kc.check_gemm_tn(sim.device, Mc=2048 + 32, P=1024, Q=256, seed=23)


def test_gemm_tn_pair_placements_sim(sim, monkeypatch):
    """Seeded sweep over the grouped launch's host-side placement: chip sizes that do / do not split into 8 'XCDs', reserved
    compute units, one or two problems of 1-3 x 1-2 tiles, ragged contraction lengths - every (problem, tile, slice) must be
    covered exactly once whatever is left of an XCD's slots."""
    import random
    from ccd_amd import ops
    rng = random.Random(7)
    for case in range(8):
        cus = rng.choice([1, 3, 4, 8, 9, 16, 17, 24])
        reserve = rng.choice([0, 0, 1])
        monkeypatch.setenv("CCD_SIM_CUS", str(cus))
        shape1 = (384 * rng.choice([1, 2, 3]), 192 * rng.choice([1, 2]))
        shape2 = (384 * rng.choice([1, 2]), 192 * rng.choice([1, 2]))
        mc = 2048 + 32 * rng.randrange(0, 8)
        with ops.policy(cu_reserve=min(reserve, cus - 1), gemm_tn384_min_tiles=1):
            if rng.random() < 0.7:
                kc.check_gemm_tn_pair(sim.device, mc, shape1, shape2, seed=30 + case)
            else:
                kc.check_gemm_tn(sim.device, Mc=mc, P=shape1[0], Q=shape1[1], seed=30 + case)


def test_gemm_tn384_sim(sim, monkeypatch):
    """gemm_tn384.h (LDS-DMA image + transposing LDS reads): one workgroup with 67 stages; 4 ragged slices of 17 / 16 stages;
    two tiles per group and two groups; 16 workgroups spread over 8 'XCDs' (2 groups of one tile each, 4 stages)."""
    kc.check_gemm_tn(sim.device, Mc=2048 + 96, P=384, Q=192)
    monkeypatch.setenv("CCD_SIM_CUS", "4")
    kc.check_gemm_tn(sim.device, Mc=2048 + 96, P=384, Q=192, seed=2)
    kc.check_gemm_tn(sim.device, Mc=2048, P=384, Q=384, seed=3)
    monkeypatch.setenv("CCD_SIM_CUS", "16")
    kc.check_gemm_tn(sim.device, Mc=2048, P=384, Q=192, seed=4)


def test_rowproj_sim(sim, monkeypatch):
    """rowproj.h on the CPU executor: one and several tiles per workgroup (the weight ring runs across tiles), ragged rows,
    K = 384 with two row blocks per wave and K = 512 with one."""
    monkeypatch.setenv("CCD_SIM_CUS", "2")
    kc.check_rowproj(sim.device, M=300, N=128, K=384)            # 2 tiles of 256 rows on 2 workgroups
    kc.check_rowproj(sim.device, M=256 * 3 + 40, N=192, K=384, seed=3)   # 4 tiles on 2 workgroups: 2 each
    kc.check_rowproj(sim.device, M=200, N=64, K=512, seed=4, strided=False)
    from ccd_amd import ops
    with ops.policy(rowproj_rb=1):                # 128-row tiles at K = 384 (what a half-empty chip gets)
        kc.check_rowproj(sim.device, M=128 * 3 + 40, N=192, K=384, seed=5)
    with ops.policy(rowproj_rb=2):
        kc.check_rowproj(sim.device, M=300, N=128, K=384, seed=6)


def test_layernorm_sim(sim):
    kc.check_layernorm(sim.device, rows=37, E=192)
    kc.check_layernorm(sim.device, rows=9, E=384)
    kc.check_layernorm(sim.device, rows=37, E=192, g16=True)      # (round 6) the gradient stream as a bf16 tensor


def test_attention_sim(sim):
    from ccd_amd import ops
    kc.check_attention(sim.device, views=1, heads=2)
    with ops.policy(attn_tr=0):                              # dK / dV on the four register-staged images
        kc.check_attention(sim.device, views=1, heads=2)
    kc.check_attention(sim.device, views=3, heads=1, seed=4)
    with ops.policy(attn_onepass=0):             # the dQ + dK/dV pair (rounds 1-3)
        kc.check_attention(sim.device, views=2, heads=2, seed=6)
    with ops.policy(attn_onepass=1):             # round 4: one pass, five products, dS exchanged through LDS (attention_bwd1.h)
        kc.check_attention(sim.device, views=1, heads=2)
        kc.check_attention(sim.device, views=6, heads=3, seed=5)      # 18 blocks on 4 workgroups: heads change inside a workgroup's walk


def test_gemm_dynamic_rows_sim(sim):
    import torch
    from ccd_amd import ops
    g = torch.Generator().manual_seed(0)
    a = kc.rnd((300, 64), g).to(kc.BF); b = kc.rnd((72, 64), g).to(kc.BF)
    d_rows = torch.tensor([75], dtype=torch.int32)
    out = torch.full((300, 72), 7.0)
    ops.gemm_nt(a, b, epilogue=ops.EPI_F32, out=out, d_rows=d_rows, rows_mul=2)
    kc.close(out[:150], a[:150].float() @ b.float().t(), 1e-4, 1e-4, "dyn/nt")
    assert (out[150:] == 7.0).all()
    c = kc.rnd((300, 40), g).to(kc.BF)
    acc = torch.zeros(64, 40)
    ops.gemm_tn(a, c, acc, d_rows=d_rows, rows_mul=2, splits=2)
    kc.close(acc, a[:150].float().t() @ c[:150].float(), 1e-4, 1e-2, "dyn/tn")


def test_ccl_sim(sim):
    kc.check_ccl(sim.device)


def test_seg_to_mask_sim(sim):
    kc.check_seg_to_mask(sim.device)
    kc.check_predicted_mask_chain(sim.device)


def test_warp_sim(sim):
    kc.check_warp(sim.device)


def test_region_sim(sim):
    kc.check_region(sim.device)
    kc.check_region_adjacent_planes(sim.device)


def test_patch_embed_sim(sim):
    kc.check_patch_embed(sim.device)


def test_small_ops_sim(sim):
    kc.check_small_ops(sim.device)


def test_head_pieces_sim(sim):
    kc.check_head_pieces(sim.device)


def test_head_loss_sim(sim, monkeypatch):
    """headloss.h on the CPU executor: 8 workgroups (one per 'XCD'), each walking several (split, row tile) items with the weight
    ring running across them; rows that straddle M inside a tile, a ragged last tile; 16 workgroups; one chunk and two per split."""
    monkeypatch.setenv("CCD_SIM_CUS", "8")
    kc.check_head_loss(sim.device, M=11, K=512)                 # 1 row tile, 8 splits of one chunk
    kc.check_head_loss(sim.device, M=75, K=1024, seed=22)       # 2 row tiles (150 rows), 8 splits of two chunks
    monkeypatch.setenv("CCD_SIM_CUS", "16")
    kc.check_head_loss(sim.device, M=70, K=1536, seed=23, spare=0)    # 24 chunks: 24 splits of one chunk, 2 row tiles, 16 workgroups


def test_dino_loss_sim(sim):
    kc.check_dino_loss(sim.device)


def test_seg_loss_sim(sim):
    kc.check_seg_loss(sim.device)


def test_optimizer_sim(sim):
    kc.check_optimizer(sim.device)


def test_mlp_bwd_fused_sim(sim, monkeypatch):
    """mlp_bwd.h on the CPU executor: ragged tiles, several tiles per workgroup (the ring and the u images run across them), both widths."""
    monkeypatch.setenv("CCD_SIM_CUS", "2")
    kc.check_mlp_bwd_fused(sim.device, M=200, E=384, H=128, rps=8)
    kc.check_mlp_bwd_fused(sim.device, M=128 * 4 + 40, E=256, H=256, rps=128, seed=5)      # 5 tiles on 2 workgroups
    kc.check_mlp_bwd_fused(sim.device, M=128 * 3, E=384, H=384, rps=64, seed=6)            # 3 tiles: one workgroup walks two


def test_conv_pieces_sim(sim):
    kc.check_conv_pieces(sim.device)


def test_seghead_sim(sim):
    kc.check_seghead(sim.device, images=1, E=64)


def test_gemm256_sim(sim):
    """The 256x256 LDS-DMA kernel, forced onto small ragged problems (several tiles per workgroup, edge tiles)."""
    from ccd_amd import ops
    with ops.policy(gemm_256_min_m=1, gemm_256_min_n=1):
        kc.check_gemm_nt(sim.device, M=300, N=264, K=128)
        kc.check_gemm_dynamic_rows(sim.device, M=600, N=264, K=64, live=75)
        with ops.policy(gemm_256_deep=1, gemm_256_f32=1):       # deep-prefetch variant (BK = 32, four buffers)
            kc.check_gemm_nt(sim.device, M=300, N=264, K=128)
            kc.check_gemm_nt(sim.device, M=260, N=256, K=64)
            kc.check_gemm_dynamic_rows(sim.device, M=600, N=264, K=192, live=75)
    # the 256x128 variant (all epilogues, incl. fp32 residual / fp32 stores)
    with ops.policy(gemm_256=2, gemm_256_min_m=1, gemm_256_min_n=1000000):
        kc.check_gemm_nt(sim.device, M=300, N=136, K=128)
        kc.check_gemm_dynamic_rows(sim.device, M=600, N=136, K=64, live=75)


def test_gemm_row384_sim(sim):
    """Full-row kernel (128 x 384 tile), forced onto small ragged problems."""
    from ccd_amd import ops
    with ops.policy(gemm_256_min_m=1, gemm_row384=2):
        kc.check_gemm_nt(sim.device, M=300, N=136, K=128)
        kc.check_gemm_nt(sim.device, M=140, N=384, K=192)
        kc.check_gemm_dynamic_rows(sim.device, M=600, N=264, K=64, live=75)


def test_policy_table_sim(sim):
    """ccd_policy_set / _get: known keys round-trip, unknown keys are rejected, the context manager restores."""
    from ccd_amd import ops
    before = ops.policy_get("gemm_256_min_m")
    with ops.policy(gemm_256_min_m=7):
        assert ops.policy_get("gemm_256_min_m") == 7
    assert ops.policy_get("gemm_256_min_m") == before
    with pytest.raises(RuntimeError):
        ops.policy_set("no_such_key", 1)


# CCD_SIM_DMA=late: the executor delivers LDS-DMA data only when a counted wait retires it (hipsim.h)
LATE_DMA = os.environ.get("CCD_SIM_DMA", "").startswith("l")


def test_gemm_lnbwd_sim(sim):
    from ccd_amd import ops
    kc.check_gemm_lnbwd(sim.device, M=300, N=384, K=384)      # rowgemm.h (N in {128, 256, 384, 512})
    kc.check_gemm_lnbwd(sim.device, M=1100, N=384, K=576, seed=34)      # 9 row tiles on 4 workgroups: the rings run across tiles
    kc.check_gemm_lnbwd(sim.device, M=200, N=128, K=256)
    kc.check_gemm_lnbwd(sim.device, M=260, N=256, K=256)
    kc.check_gemm_lnbwd(sim.device, M=130, N=512, K=128)
    kc.check_gemm_lnbwd(sim.device, M=70, N=384, K=768)
    kc.check_gemm_lnbwd(sim.device, M=300, N=384, K=128)      # gemm_row384.h
    kc.check_gemm_lnbwd(sim.device, M=140, N=192, K=64)
    # round 6: the gradient stream as a bf16 tensor (row-owner kernels: ADMA and register-loaded activation rows, N = 128 / 256)
    kc.check_gemm_lnbwd(sim.device, M=300, N=384, K=384, g16=True)
    kc.check_gemm_lnbwd(sim.device, M=1100, N=384, K=576, seed=34, g16=True)
    kc.check_gemm_lnbwd(sim.device, M=200, N=128, K=256, g16=True)
    kc.check_gemm_lnbwd(sim.device, M=260, N=256, K=256, g16=True)
    with ops.policy(rowgemm_adma=0):
        kc.check_gemm_lnbwd(sim.device, M=300, N=384, K=384, seed=35, g16=True)
    # round 6: a tap's LayerNorm backward in the same epilogue (both row loaders)
    kc.check_gemm_lnbwd(sim.device, M=300, N=384, K=384, seed=37, g16=True, tap=True)
    kc.check_gemm_lnbwd(sim.device, M=1100, N=384, K=576, seed=38, g16=True, tap=True)
    with ops.policy(rowgemm_adma=0):
        kc.check_gemm_lnbwd(sim.device, M=300, N=384, K=384, seed=39, g16=True, tap=True)


def test_gemm_resid_ln_sim(sim):
    from ccd_amd import ops
    with ops.policy(rowgemm=2):                                   # rowgemm.h forced (default only at N = 512)
        kc.check_gemm_resid_ln(sim.device, M=300, N=384, K=192)
        kc.check_gemm_resid_ln(sim.device, M=200, N=128, K=256)
    kc.check_gemm_resid_ln(sim.device, M=130, N=512, K=128)
    kc.check_gemm_resid_ln(sim.device, M=300, N=384, K=128)      # gemm_row384.h
    kc.check_gemm_resid_ln(sim.device, M=140, N=192, K=64)


def test_decoder_pieces_sim(sim):
    kc.check_decoder_pieces(sim.device)


def test_decoder_abi_rejects_unsupported_arguments(sim):
    """Argument errors of the finetune-path entry points come back as error codes (-> RuntimeError), never as launches."""
    import torch
    from ccd_amd import ops
    bf = torch.bfloat16
    q = torch.zeros((2 * 40, 128), dtype=bf)
    with pytest.raises(RuntimeError, match="unsupported shape"):          # 40 queries > 32
        ops.dec_attn_fwd(q, q, q, 2, 2, 40, 40, 0.125)
    k = torch.zeros((2 * 300, 128), dtype=bf)
    with pytest.raises(RuntimeError, match="unsupported shape"):          # 300 keys > 256
        ops.dec_attn_fwd(q[:2 * 25], k, k, 2, 2, 25, 300, 0.125)
    with pytest.raises(RuntimeError, match="invalid argument"):           # dropout probability outside [0, 1)
        ops.dropout(torch.zeros(64), 1.0, 1)
    with pytest.raises(RuntimeError, match="unsupported shape"):          # element count not a multiple of 4
        ops.dropout(torch.zeros(6), 0.1, 1)
    logits = torch.zeros((50, 200))
    with pytest.raises(RuntimeError, match="unsupported shape"):          # more than 128 classes
        ops.tf_loss_fwd(logits, 200, torch.zeros((2, 25), dtype=torch.int64), 92)
    # empty batches are a no-op, not an error
    e = torch.zeros((0, 128), dtype=bf)
    out, lse, _ = ops.dec_attn_fwd(e, e, e, 0, 2, 25, 25, 0.125)
    assert out.shape == (0, 128)
    assert ops.dropout(torch.zeros(0), 0.1, 1).numel() == 0


def test_matvec_bf16_sim(sim):
    kc.check_matvec_bf16(sim.device)
    kc.check_matvec_bf16(sim.device, K=70, D=512, seed=52)


def test_proj_mlp_fused_sim(sim):
    """mlp_fused.h with the projection + residual + LayerNorm-2 prologue (PROJ): ring seeks over dropped branches, both epilogues,
    several tiles per workgroup (1 CU), with and without the tensors saved for the backward pass."""
    kc.check_proj_mlp_fused(sim.device, M=128 * 5 + 40, E=128, H=256, rps=128)
    kc.check_proj_mlp_fused(sim.device, M=300, E=384, H=128, rps=128, save=False)
    kc.check_proj_mlp_fused(sim.device, M=200, E=384, H=320, rps=256, seed=43)
    kc.check_proj_mlp_fused(sim.device, M=130, E=256, H=128, rps=128, seed=44, drops=False)


def test_mlp_fused_sim(sim):
    """Ragged last tile, several tiles per workgroup (1 CU), a dropped sample, both instantiations of E."""
    kc.check_mlp_fused(sim.device, M=300, E=128, H=256, rps=128)
    kc.check_mlp_fused(sim.device, M=200, E=384, H=128, rps=8, store_u=False)     # per-row DropPath scales
    kc.check_mlp_fused(sim.device, M=130, E=512, H=128, rps=8)                     # 3-slot ring (vit_base)
    kc.check_mlp_fused(sim.device, M=140, E=384, H=320, rps=8)                     # 5 hidden chunks at E = 384


def test_kmeans2_mask_sim(sim):
    kc.check_kmeans2_mask(sim.device, extra=4)


def test_multi_launch_helpers_sim(sim):
    kc.check_multi_launch_helpers(sim.device)


def test_cls_tail_sim(sim):
    kc.check_cls_tail(sim.device, images=1)


def test_augment_views_sim(sim):
    kc.check_augment_views(sim.device, H=16, W=40)


def test_cu_reserve_window_sim(sim, monkeypatch):
    """cu_reserve_window >= 0: only the launches right behind a bucket launch (cu_reserve_left, re-armed by the gradient reducer)
    leave compute units free; results do not depend on the grid either way."""
    from ccd_amd import ops
    monkeypatch.setenv("CCD_SIM_CUS", "4")
    with ops.policy(cu_reserve=2, cu_reserve_window=3, cu_reserve_left=2):
        kc.check_gemm_nt(sim.device, M=200, N=136, K=128)          # several persistent-grid launches: the counter runs out
        assert ops.policy_get("cu_reserve_left") == 0
        kc.check_gemm_tn(sim.device, Mc=300, P=136, Q=72, splits=3)
        ops.policy_set("cu_reserve_left", 1)
        kc.check_gemm_tn(sim.device, Mc=2048, P=384, Q=192, seed=9)   # one gemm_tn384 launch on 2 of the 4 CUs, then the rest on 4
        assert ops.policy_get("cu_reserve_left") == 0


@pytest.mark.skipif(LATE_DMA, reason="this IS the late-DMA run")
def test_kernels_under_late_dma_model():
    """The whole file once more with the executor delivering LDS-DMA data as LATE as the hardware may (only when a counted wait of
    the issuing wave retires the request, hipsim.h): a fragment read in front of a sufficient `vmcnt` wait reads stale LDS.  The
    default run delivers them as EARLY as possible (buffer-reuse hazards); asynchronous-copy bookkeeping is invisible to a
    synchronous executor otherwise - two such bugs of round 2 only showed on the GPU."""
    import subprocess
    import sys
    env = dict(os.environ, CCD_SIM_DMA="late")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
