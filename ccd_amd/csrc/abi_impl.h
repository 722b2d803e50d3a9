// abi_impl.h - the extern "C" entry points declared in include/ccd_hip.h: argument checks + launch geometry.
// Included by ccd_hip.hip (product, after prelude_hip.h) and by tests/hipsim/ccd_sim.cpp (CPU SIMT executor,
// test infrastructure) - hence launches go through CCD_LAUNCH and the ccd_rt_* helpers of the prelude.
#pragma once

#include "../../include/ccd_hip.h"
#include "kernels/common.h"
#include "kernels/gemm.h"
#include "kernels/gemm256.h"
#include "kernels/gemm_row384.h"
#include "kernels/mlp_fused.h"
#include "kernels/rowgemm.h"
#include "kernels/mlp_bwd.h"
#include "kernels/rowproj.h"
#include "kernels/attention_fwd.h"
#include "kernels/attention_bwd.h"
#include "kernels/attention_bwd1.h"
#include "kernels/gemm_tn384.h"
#include "kernels/layernorm.h"
#include "kernels/charmap.h"
#include "kernels/datapipe.h"
#include "kernels/embed.h"
#include "kernels/head.h"
#include "kernels/loss.h"
#include "kernels/headloss.h"
#include "kernels/optim.h"
#include "kernels/conv.h"
#include "kernels/cls_tail.h"
#include "kernels/decoder.h"
#include "kernels/decoder_xattn.h"

#define CCD_CHECK(cond, code) \
    do {                      \
        if (!(cond)) return (code); \
    } while (0)
#define CCD_ALIGNED16(p) ((((uintptr_t)(p)) & 15u) == 0)
#define CCD_MAX_OPERAND_BYTES 0x7ffffff0L

static int ccd_grid_cus(long tiles = 0);
// persistent grid: one workgroup per resident slot (2 per CU), never more than there are work items
static int ccd_gemm_grid(ccd::GemmParams& p, int tiles, int splits) {
    p.work_items = tiles * splits;
    const int cap = 2 * ccd_grid_cus((p.work_items + 1) / 2);      // (split counts are sized for the whole chip: no extra round for 16 left-over items)
    return p.work_items < cap ? p.work_items : cap;
}

template <bool TN>
static int ccd_launch_gemm(ccd::GemmParams p, int epilogue, int splits, void* stream) {
    const int tiles = ((p.M + ccd::GEMM_BM - 1) / ccd::GEMM_BM) * ((p.N + ccd::GEMM_BN - 1) / ccd::GEMM_BN);
    const dim3 grid(ccd_gemm_grid(p, tiles, splits)), block(256);
    const size_t smem = ccd::GEMM_SMEM_BYTES;
    switch (epilogue) {
        case CCD_EPI_BF16: CCD_LAUNCH((ccd::gemm_bf16_kernel<TN, ccd::EPI_BF16>), grid, block, smem, stream, p); break;
        case CCD_EPI_GELU: CCD_LAUNCH((ccd::gemm_bf16_kernel<TN, ccd::EPI_GELU>), grid, block, smem, stream, p); break;
        case CCD_EPI_RESID: CCD_LAUNCH((ccd::gemm_bf16_kernel<TN, ccd::EPI_RESID>), grid, block, smem, stream, p); break;
        case CCD_EPI_F32: CCD_LAUNCH((ccd::gemm_bf16_kernel<TN, ccd::EPI_F32>), grid, block, smem, stream, p); break;
        case CCD_EPI_ATOMIC: CCD_LAUNCH((ccd::gemm_bf16_kernel<TN, ccd::EPI_ATOMIC>), grid, block, smem, stream, p); break;
        case CCD_EPI_DGELU: CCD_LAUNCH((ccd::gemm_bf16_kernel<TN, ccd::EPI_DGELU>), grid, block, smem, stream, p); break;
        default: return CCD_EINVAL;
    }
    return ccd_rt_last_error();
}

// blocks of 256 threads for a grid-stride elementwise kernel over `total` 16-byte chunks, `per_thread` chunks per loop trip:
// enough to give every thread one trip, at most 16 blocks per CU (the stride stays a multiple of 256)
static unsigned ccd_stream_blocks(long total, int per_thread) {
    long blocks = (total + 256L * per_thread - 1) / (256L * per_thread);
    const long cap = 16L * ccd_rt_num_cus();
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}
// launch geometry of the column reductions (colsum_bf16, bn_relu_bwd_reduce): cgn = 2^cgn_log2 column groups of 8
// per 1024-thread block (<= 32), every block streams >= 128 KiB (1 MiB left a 17-MB tensor on 32 of the 256 CUs: 60 us where
// 10 do), at most 2 blocks per CU
static void ccd_reduce_geometry(long rows, int N, int* cgn_log2, int* col_blocks, int* rows_per_block, int* row_blocks) {
    int lg = 0;
    while (lg < 5 && (8 << lg) < N) ++lg;
    const int cgn = 1 << lg, rln = ccd::COLSUM_THREADS >> lg;
    *cgn_log2 = lg;
    *col_blocks = (N + 8 * cgn - 1) / (8 * cgn);
    long rb = (rows * (long)(16 * cgn) + (1L << 17) - 1) / (1L << 17);           // bytes per block-column / 128 KiB
    const long cap = (2L * ccd_rt_num_cus() + *col_blocks - 1) / *col_blocks;
    if (rb > cap) rb = cap;
    if (rb < 1) rb = 1;
    long rpb = (rows + rb - 1) / rb;
    rpb = ((rpb + rln - 1) / rln) * rln;
    *rows_per_block = (int)rpb;
    *row_blocks = (int)((rows + rpb - 1) / rpb);
}
// Kernel-selection policy: a small table of integers parsed ONCE from the environment (CCD_<KEY in upper case>) and
// changeable at run time through ccd_policy_set (include/ccd_hip.h) - no getenv on the launch path.
struct CcdPolicy {
    int gemm_256 = 1;           // 0 off, 1 = 256x256 tiles for N >= gemm_256_min_n bf16-output products, 2 = also 256x128 tiles
    int gemm_256_min_m = 2048, gemm_256_min_n = 384;
    int gemm_256_f32 = 0;       // 256-row kernels also for the fp32 / residual epilogues
    int gemm_256_deep = 0;      // BK = 32 x 4 buffers (three k-steps of DMA in flight) instead of BK = 64 x 2
    int gemm_row384 = 0;        // 1 = full-row kernel for N <= 384 residual / fp32 epilogues, 2 = bf16 too
    int rowproj = 1;            // K = 384 / 512 bf16 projections (qkv, proj data gradient) with the activation rows resident in registers (rowproj.h); 0 = gemm256.h / gemm.h
    int rowproj_min_m = 16384;  // ... from this many rows on (a workgroup tile is 256 rows: below ~64 tiles the 128-row kernels fill the chip better)
    int rowproj_rb = 0;         // row blocks per wave at K = 384: 2 (256-row tiles), 1 (128-row tiles), 0 = 1 where 256-row tiles leave compute units idle (64 images per GPU: 32 768 rows = 128 tiles)
    int rowgemm = 1;            // row-owner kernels (rowgemm.h) for the N in {128, 256, 384} row-wise epilogues; 0 = gemm_row384.h
    int ln_bwd_bpc = 5;         // LayerNorm backward: blocks per CU (one resident wave; more blocks = more dgamma/dbeta atomics)
    int dec_attn_simt = 0;      // decoder attention: force the general SIMT kernels
    int attn_tr = 1;            // attention backward dK/dV: double-buffered LDS-DMA row images + ds_read_b64_tr_b16 (0 = four register-staged images)
    int attn_onepass = 1;       // attention backward: ONE kernel with five products, dS exchanged through LDS (attention_bwd1.h); 0 = the dQ + dK/dV pair
    int attn_skew = 0;          // attention backward: waves 4..7 start each block ~skew * 64 cycles late (lab; no effect once clocks are warm)
    int gemm_tn384 = 1;         // weight gradients with P % 384 == 0, Q % 192 == 0: XCD-grouped 384x192 LDS-DMA kernel (gemm_tn384.h); 0 = 128-square kernel, 2 = never as a pair
    int gemm_tn384_geom = 0;    // its workgroup: 0 = 384x192 tile, 8 waves, one per CU; 2 = 0 + 512x128 tiles for the shapes 384x192 does not divide (E = 512)
    int gemm_tn384_min_tiles = 6;   // ... for a SINGLE product only from this many tiles on (proj, 2 tiles = 128 slices: the atomic epilogue dominates)
    int cu_reserve = 0;         // compute units the persistent grids leave free (set while an RCCL gradient reducer is attached)
    int cu_reserve_window = -1; // -1: every launch leaves them free; N >= 0: only the next `cu_reserve_left` launches do (the reducer
    int cu_reserve_left = 0;    // re-arms it with N whenever it starts a bucket's all-reduce: the kernels that run beside the collective)
    int rowgemm_adma = 1;       // row-owner products at N = 384: the activation rows by LDS-DMA into per-wave images (rowgemm.h, ADMA); 0 = row-per-lane register loads
    int tn_ws = 1;              // paired weight gradients: per-slice partial stores + a reduction pass when the caller lends a workspace (0 = fp32 atomics)
    int lab = 0;                // scratch switch for kernel experiments (tools/*_lab.py); 0 in production
};
struct CcdPolicyKey { const char* name; int CcdPolicy::*field; };
static const CcdPolicyKey ccd_policy_keys[] = {
    {"gemm_256", &CcdPolicy::gemm_256}, {"gemm_256_min_m", &CcdPolicy::gemm_256_min_m},
    {"gemm_256_min_n", &CcdPolicy::gemm_256_min_n}, {"gemm_256_f32", &CcdPolicy::gemm_256_f32},
    {"gemm_256_deep", &CcdPolicy::gemm_256_deep}, {"gemm_row384", &CcdPolicy::gemm_row384},
    {"rowproj", &CcdPolicy::rowproj}, {"rowproj_min_m", &CcdPolicy::rowproj_min_m}, {"rowproj_rb", &CcdPolicy::rowproj_rb},
    {"rowgemm", &CcdPolicy::rowgemm}, {"ln_bwd_bpc", &CcdPolicy::ln_bwd_bpc}, {"dec_attn_simt", &CcdPolicy::dec_attn_simt}, {"attn_onepass", &CcdPolicy::attn_onepass}, {"attn_skew", &CcdPolicy::attn_skew}, {"attn_tr", &CcdPolicy::attn_tr}, {"gemm_tn384", &CcdPolicy::gemm_tn384}, {"gemm_tn384_min_tiles", &CcdPolicy::gemm_tn384_min_tiles}, {"gemm_tn384_geom", &CcdPolicy::gemm_tn384_geom}, {"cu_reserve", &CcdPolicy::cu_reserve}, {"cu_reserve_window", &CcdPolicy::cu_reserve_window}, {"cu_reserve_left", &CcdPolicy::cu_reserve_left}, {"rowgemm_adma", &CcdPolicy::rowgemm_adma}, {"tn_ws", &CcdPolicy::tn_ws}, {"lab", &CcdPolicy::lab}};
static CcdPolicy& ccd_policy() {
    static CcdPolicy pol = [] {
        CcdPolicy q;
        for (const CcdPolicyKey& k : ccd_policy_keys) {
            char env[64] = "CCD_";
            size_t n = 4;
            for (const char* c = k.name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
            env[n] = 0;
            if (const char* v = getenv(env)) q.*(k.field) = atoi(v);
        }
        return q;
    }();
    return pol;
}
// compute units a persistent grid may occupy: all of them, minus the ones reserved for concurrently running RCCL kernels
// Reserving CUs is expensive for the row-owner kernels (131072 rows = 512 tiles of 256 rows: 2 rounds on 256 CUs, 3 on 248 - the
// fused MLP, the residual + LayerNorm product and the LayerNorm-backward product ran 16-20 % longer), and a bucket's all-reduce is in
// flight for well under a millisecond: with cu_reserve_window >= 0 only the launches right behind a bucket launch leave CUs free.
// `tiles` (row-owner launches, one workgroup per CU, a few large tiles each): a launch whose tiles fill the whole chip in r rounds and
// would need r + 1 on the reduced grid (512 tiles of 256 rows: 2 rounds on 256 CUs, 3 on 248) takes every CU - the collective's
// workgroups then get in when this launch's first workgroups retire, at most one kernel (< 0.7 ms) late, instead of every such launch
// paying half its run time for CUs the collective needs for a fraction of it.  The launch still uses up one slot of the window.
static int ccd_grid_cus(long tiles) {
    CcdPolicy& pol = ccd_policy();
    int reserve = pol.cu_reserve;
    if (reserve > 0 && pol.cu_reserve_window >= 0) {
        if (pol.cu_reserve_left > 0) --pol.cu_reserve_left;
        else reserve = 0;
    }
    const int all = ccd_rt_num_cus();
    if (reserve > 0 && tiles > 0 && 8 * reserve <= all) {        // (a reserve of most of the chip is a test forcing a small grid)
        const long full = (tiles + all - 1) / all, less = (tiles + (all - reserve) - 1) / (all - reserve);
        if (less > full && full <= 4) reserve = 0;
    }
    const int cus = all - reserve;
    return cus > 1 ? cus : 1;
}
// row-owner projection (rowproj.h): out bf16 = A . B^T + bias, K in {384, 512}, N % 64 == 0
static bool ccd_rowproj_takes(const ccd::GemmParams& p, int epilogue) {
    const CcdPolicy& pol = ccd_policy();
    return pol.rowproj && epilogue == CCD_EPI_BF16 && (p.K == 384 || p.K == 512) && p.N % 64 == 0 && p.N <= 4096 &&
           p.M >= pol.rowproj_min_m && !p.d_rows && !p.colsum && p.alpha == 1.0f && p.ldc % 8 == 0 &&
           ccd::rp_smem_bytes(p.K, p.N) <= 160 * 1024 &&
           ((long)p.M + 512) * p.lda * 2 < CCD_MAX_OPERAND_BYTES && ((long)p.M + 512) * p.ldc * 2 < CCD_MAX_OPERAND_BYTES;
}
static int ccd_launch_rowproj(const ccd::GemmParams& p, void* stream) {
    ccd::RowProjParams q;
    q.a = reinterpret_cast<const ccd::bf16_t*>(p.A); q.lda = p.lda; q.w = reinterpret_cast<const ccd::bf16_t*>(p.B); q.ldw = p.ldb;
    q.bias = p.bias; q.out = reinterpret_cast<ccd::bf16_t*>(p.C); q.ldc = p.ldc; q.M = p.M; q.N = p.N;
    const int cus = ccd_grid_cus((p.M + ccd::rp_rows(p.K == 384 ? 2 : 1) - 1) / ccd::rp_rows(p.K == 384 ? 2 : 1)), smem = ccd::rp_smem_bytes(p.K, p.N);
    if (p.K == 384) {
        const int tiles = (p.M + ccd::rp_rows(2) - 1) / ccd::rp_rows(2), rb = ccd_policy().rowproj_rb;
        if (rb == 1 || (rb == 0 && tiles < cus)) {       // (round 5) the chip is not full of 256-row tiles: half the rows per workgroup
            const int tiles1 = (p.M + ccd::rp_rows(1) - 1) / ccd::rp_rows(1);
            CCD_LAUNCH((ccd::rowproj_kernel<384, 1>), dim3(tiles1 < cus ? tiles1 : cus), dim3(ccd::RP_THREADS), smem, stream, q);
        } else {
            CCD_LAUNCH((ccd::rowproj_kernel<384, 2>), dim3(tiles < cus ? tiles : cus), dim3(ccd::RP_THREADS), smem, stream, q);
        }
    } else {
        const int tiles = (p.M + ccd::rp_rows(1) - 1) / ccd::rp_rows(1);
        CCD_LAUNCH((ccd::rowproj_kernel<512, 1>), dim3(tiles < cus ? tiles : cus), dim3(ccd::RP_THREADS), smem, stream, q);
    }
    return ccd_rt_last_error();
}
// 256x256-tile LDS-DMA kernel for the large-M products (gemm256.h): one workgroup per CU
template <int BN, bool DEEP = false>
static int ccd_launch_gemm256(const ccd::GemmParams& p, int epilogue, void* stream) {
    const int tiles = ((p.M + ccd::G256_BM - 1) / ccd::G256_BM) * ((p.N + BN - 1) / BN);
    const int cus = ccd_grid_cus();
    const dim3 grid(tiles < cus ? tiles : cus), block(ccd::G256_THREADS);
    const size_t smem = ccd::g256_smem_bytes(p.N, p.colsum != nullptr);
    switch (epilogue) {
        case CCD_EPI_BF16: CCD_LAUNCH((ccd::gemm256_kernel<ccd::EPI_BF16, BN, DEEP>), grid, block, smem, stream, p); break;
        case CCD_EPI_GELU: CCD_LAUNCH((ccd::gemm256_kernel<ccd::EPI_GELU, BN, DEEP>), grid, block, smem, stream, p); break;
        case CCD_EPI_RESID: CCD_LAUNCH((ccd::gemm256_kernel<ccd::EPI_RESID, BN, DEEP>), grid, block, smem, stream, p); break;
        case CCD_EPI_F32: CCD_LAUNCH((ccd::gemm256_kernel<ccd::EPI_F32, BN, DEEP>), grid, block, smem, stream, p); break;
        case CCD_EPI_DGELU: CCD_LAUNCH((ccd::gemm256_kernel<ccd::EPI_DGELU, BN, DEEP>), grid, block, smem, stream, p); break;
        default: return CCD_EINVAL;
    }
    return ccd_rt_last_error();
}
// full-row kernel for N <= 384 (gemm_row384.h): one workgroup per CU
// `cus`: the caller's ccd_grid_cus() when it already asked (one launch consumes ONE slot of the cu_reserve window), -1 otherwise
static int ccd_launch_gemm_row384(const ccd::GemmParams& p, int epilogue, void* stream, int cus = -1) {
    const int tiles = (p.M + ccd::GR_BM - 1) / ccd::GR_BM;
    if (cus < 0) cus = ccd_grid_cus(tiles);
    const dim3 grid(tiles < cus ? tiles : cus), block(ccd::GR_THREADS);
    const size_t smem = ccd::GR_SMEM_BYTES;
    switch (epilogue) {
        case CCD_EPI_BF16: CCD_LAUNCH((ccd::gemm_row384_kernel<ccd::EPI_BF16>), grid, block, smem, stream, p); break;
        case CCD_EPI_RESID: CCD_LAUNCH((ccd::gemm_row384_kernel<ccd::EPI_RESID>), grid, block, smem, stream, p); break;
        case CCD_EPI_F32: CCD_LAUNCH((ccd::gemm_row384_kernel<ccd::EPI_F32>), grid, block, smem, stream, p); break;
        case 7: CCD_LAUNCH((ccd::gemm_row384_kernel<ccd::EPI_RESID_LN>), grid, block, smem, stream, p); break;
        case 8: CCD_LAUNCH((ccd::gemm_row384_kernel<ccd::EPI_LNBWD>), grid, block, smem, stream, p); break;
        default: return CCD_EINVAL;
    }
    return ccd_rt_last_error();
}

// gemm_tn384.h launch for one workgroup geometry (see ccd_launch_tn384 below)
template <int WM, int WN, int STAGES, int TI, int TJ>
static int ccd_launch_tn384_geom(ccd::GemmParams& p, int Mc, float* ws, long ws_floats, void* stream) {
    using G = ccd::Tn3Geom<WM, WN, STAGES, TI, TJ>;
    const int per_cu = 8 / G::WAVES;
    const int t1 = (p.M / G::TP) * (p.N / G::TQ), t2 = (p.M2 / G::TP) * (p.N2 / G::TQ), slots = per_cu * ccd_grid_cus();
    const int xcds = slots >= 8 * (t1 > t2 ? t1 : t2) ? 8 : 1;
    const int spx = slots / xcds;
    // groups per XCD: always the problem with fewer groups so far (ties: the larger group first) while it fits - both problems
    // end up with (nearly) the same number of slices, i.e. rows per workgroup, whatever is left of an XCD's slots
    int s1 = 0, s2 = 0;
    unsigned long long units1 = 0, units2 = 0;
    for (int x = 0; x < xcds; ++x) {
        int rem = spx, u1 = 0, u2 = 0;
        while (true) {
            const bool first = t2 == 0 || s1 < s2 || (s1 == s2 && t1 >= t2);
            if (first && t1 <= rem && u1 < 255) { ++u1; ++s1; rem -= t1; }
            else if (!first && t2 <= rem && u2 < 255) { ++u2; ++s2; rem -= t2; }
            else if (first && t2 > 0 && s2 <= s1 && t2 <= rem && u2 < 255) { ++u2; ++s2; rem -= t2; }
            else if (!first && s1 <= s2 && t1 <= rem && u1 < 255) { ++u1; ++s1; rem -= t1; }
            else break;
        }
        units1 |= (unsigned long long)u1 << (8 * x);
        units2 |= (unsigned long long)u2 << (8 * x);
    }
    if (s1 < 1 || (t2 > 0 && s2 < 1)) return CCD_ESHAPE;
    auto rows_per = [&](int& slices) {
        int per = (Mc + slices - 1) / slices;
        per = ((per + ccd::TN3_BK - 1) / ccd::TN3_BK) * ccd::TN3_BK;
        slices = (Mc + per - 1) / per;
        return per;
    };
    p.k_per_split = rows_per(s1); p.work_items = s1;
    if (t2 > 0) { p.per2 = rows_per(s2); p.slices2 = s2; }
    p.units1 = units1; p.units2 = units2; p.m_fastest = xcds;
    // split-K partial sums by plain stores + one reduction pass when the caller lent a workspace that holds every slice's plane
    // (and there is more than one slice to add up); the fp32-atomic epilogue otherwise
    const long need1 = (long)s1 * p.M * p.N, need2 = t2 > 0 ? (long)s2 * p.M2 * p.N2 : 0;
    const bool use_ws = ws && need1 + need2 <= ws_floats && (s1 > 1 || s2 > 1) && p.ldc % 4 == 0 && (t2 == 0 || p.ldc2 % 4 == 0) &&
                        !(p.rps_shift & 1);
    p.ws = use_ws ? ws : nullptr;
    p.ws2 = use_ws ? ws + need1 : nullptr;
    CCD_LAUNCH((ccd::gemm_tn384_kernel<WM, WN, STAGES, TI, TJ>), dim3(xcds * spx), dim3(G::THREADS), G::SMEM_BYTES, stream, p);
    if (use_ws) {
        ccd::Tn3ReduceParams r;
        r.ws[0] = p.ws; r.ws[1] = p.ws2; r.C[0] = reinterpret_cast<float*>(p.C); r.C[1] = reinterpret_cast<float*>(p.C2);
        r.ldc[0] = p.ldc; r.ldc[1] = p.ldc2; r.S[0] = s1; r.S[1] = t2 > 0 ? s2 : 0; r.P[0] = p.M; r.P[1] = t2 > 0 ? p.M2 : 0;
        r.Q[0] = p.N; r.Q[1] = t2 > 0 ? p.N2 : 0; r.alpha = p.alpha;
        const long n4 = ((long)p.M * p.N + (t2 > 0 ? (long)p.M2 * p.N2 : 0)) / 4;
        CCD_LAUNCH(ccd::tn3_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, r);
    }
    return ccd_rt_last_error();
}

extern "C" {

int ccd_abi_version(void) { return 12; }   // 12: ccd_mlp_bwd_fused (gelu'(u) product + fc1 data gradient + LayerNorm-2 backward in one launch), ccd_proj_mlp_fused_gact (the forward block half also stores gelu(u)); 11: ccd_gemm_nt_lnbwd_tap_g16 (a segmentation tap's LayerNorm backward inside the qkv data-gradient product's epilogue); 10: ccd_head_loss_fwd / _bwd (last layer + distillation loss, logits never written), ccd_*_g16 (bf16 residual-gradient stream); 9: ccd_cls_tail_fwd / _bwd_reduce / _bwd_apply (BatchNorm + ReLU + classifier conv of the segmentation head fused); 8: ccd_proj_mlp_fused (proj + residual + LayerNorm-2 in front of the fused MLP), ccd_matvec_bf16; 7: ccd_gemm_tn_pair_ws (split-K workspace instead of fp32 atomics); 6: device-side momentum / DropPath seed (HIP graph of the training step); 5: ccd_mlp_fused can store gelu(u); 4: ccd_attention_bwd emits the qkv-bias gradient; 3: ccd_policy_set / _get, ccd_mlp_fused; 2: finetune-path entry points
const char* ccd_build_info(void) { return "ccd_hip gfx950 bf16-mfma abi12"; }
int ccd_policy_set(const char* key, int value) {
    CCD_CHECK(key, CCD_EINVAL);
    for (const CcdPolicyKey& k : ccd_policy_keys)
        if (!strcmp(k.name, key)) { ccd_policy().*(k.field) = value; return CCD_OK; }
    return CCD_EINVAL;
}
int ccd_policy_get(const char* key, int* value) {
    CCD_CHECK(key && value, CCD_EINVAL);
    for (const CcdPolicyKey& k : ccd_policy_keys)
        if (!strcmp(k.name, key)) { *value = ccd_policy().*(k.field); return CCD_OK; }
    return CCD_EINVAL;
}

// ----------------------------------------------------------------------------------------------- GEMM
int ccd_gemm_nt(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, int epilogue, void* C,
                long ldc, void* C2, long ldc2, const float* bias, const float* resid, long ldr,
                const float* rowscale, int rows_per_sample, const ccd_bf16* aux, long ldaux, float alpha,
                int m_fastest, const int* d_rows, int rows_mul, float* colsum, void* stream) {
    CCD_CHECK(A && B && (C || (epilogue == CCD_EPI_GELU && C2)), CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(A) && CCD_ALIGNED16(B) && CCD_ALIGNED16(C), CCD_EINVAL);
    CCD_CHECK(!colsum || epilogue == CCD_EPI_BF16 || epilogue == CCD_EPI_DGELU, CCD_EINVAL);
    if (M == 0 || N == 0) return CCD_OK;
    CCD_CHECK(M > 0 && N > 0 && K > 0, CCD_EINVAL);
    CCD_CHECK(K % 64 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, CCD_ESHAPE);
    // the loaders address operands with 32-bit byte offsets below 2 GiB (buffer loads, see gemm.h)
    CCD_CHECK(((long)M * lda + K) * 2 < CCD_MAX_OPERAND_BYTES && ((long)N * ldb + K) * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    CCD_CHECK(epilogue != CCD_EPI_GELU || (C2 && ldc2 % 8 == 0), CCD_EINVAL);
    CCD_CHECK(epilogue != CCD_EPI_RESID || (resid && ldr % 4 == 0 && rows_per_sample > 0), CCD_EINVAL);
    CCD_CHECK(epilogue != CCD_EPI_DGELU || (aux && ldaux % 8 == 0 && (!C2 || ldc2 % 8 == 0)), CCD_EINVAL);
    ccd::GemmParams p = ccd::GemmParams();
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K;
    p.C = C; p.ldc = ldc; p.C2 = C2; p.ldc2 = ldc2; p.bias = bias; p.resid = resid; p.ldr = ldr;
    p.rowscale = rowscale; p.rows_per_sample = rows_per_sample; p.aux = aux; p.ldaux = ldaux;
    p.rps_shift = -1;
    if (rows_per_sample > 0 && (rows_per_sample & (rows_per_sample - 1)) == 0)
        for (int sft = 0; sft < 31; ++sft) if ((1 << sft) == rows_per_sample) p.rps_shift = sft;
    p.k_per_split = K; p.m_fastest = m_fastest; p.alpha = alpha; p.d_rows = d_rows; p.rows_mul = rows_mul;
    p.colsum = colsum;
    // kernel choice by the policy table (defaults measured in DESIGN.md section 3; the tests change it through
    // ccd_policy_set to force small problems through the 256-row kernels)
    const CcdPolicy& pol = ccd_policy();
    if (ccd_rowproj_takes(p, epilogue)) return ccd_launch_rowproj(p, stream);
    if (pol.gemm_row384 >= 1 && N <= ccd::GR_BN && M >= pol.gemm_256_min_m &&
        (epilogue == CCD_EPI_RESID || epilogue == CCD_EPI_F32 || (epilogue == CCD_EPI_BF16 && pol.gemm_row384 >= 2)))
        return ccd_launch_gemm_row384(p, epilogue, stream);
    const bool bf16_out = epilogue == CCD_EPI_BF16 || epilogue == CCD_EPI_GELU || epilogue == CCD_EPI_DGELU;
    // fp32 / residual epilogues take the 256-row tile where the columns fill whole tiles (vit_base, N = 512: 6.6 vs 7.6 ms per
    // step); at N = 384 (one and a half tiles) the 128-row kernels are faster
    const bool f32_256 = epilogue != CCD_EPI_ATOMIC && (pol.gemm_256_f32 || N % 256 == 0);
    const bool colsum_fits = !colsum || N <= ccd::G256_MAX_COLSUM_N;      // gemm256.h keeps the column sums of every column in LDS
    // gemm256.h's gelu'(u) epilogue addresses u and both outputs with 32-bit byte offsets (buffer loads / stores)
    // (its row steps run up to 255 rows past M before the descriptor's range check drops them: the same margin as rowproj's check,
    // so that step_row * ld * 2 stays below 2^31 and an out-of-range lane's BUF_OOB + offset cannot wrap back into the buffer)
    const bool dgelu_fits = epilogue != CCD_EPI_DGELU || (((long)M + 256) * ldaux * 2 < CCD_MAX_OPERAND_BYTES &&
        ((long)M + 256) * ldc * 2 < CCD_MAX_OPERAND_BYTES && (!C2 || ((long)M + 256) * ldc2 * 2 < CCD_MAX_OPERAND_BYTES));
    if (pol.gemm_256 >= 1 && (bf16_out || f32_256) && M >= pol.gemm_256_min_m &&
        N >= pol.gemm_256_min_n && colsum_fits && dgelu_fits) {
        if (pol.gemm_256_deep) return ccd_launch_gemm256<256, true>(p, epilogue, stream);
        return ccd_launch_gemm256<256>(p, epilogue, stream);
    }
    if (pol.gemm_256 >= 2 && epilogue != CCD_EPI_ATOMIC && M >= pol.gemm_256_min_m && colsum_fits && dgelu_fits) return ccd_launch_gemm256<128>(p, epilogue, stream);
    int splits = 1;
    if (epilogue == CCD_EPI_ATOMIC && K >= 16384) {      // few output tiles, long contraction (the head's data gradient,
        p.k_per_split = 8192;                            // K = 65536: 52 live tiles): slices of 8192 accumulate by fp32 atomics
        splits = (K + 8191) / 8192;
    }
    return ccd_launch_gemm<false>(p, epilogue, splits, stream);
}

int ccd_gemm_nt_resid_ln(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, float* C, long ldc,
                         const float* bias, const float* resid, long ldr, const float* rowscale, int rows_per_sample,
                         const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* y, long ldy, float* mean,
                         float* rstd, void* stream) {
    CCD_CHECK(A && B && C && resid && ln_gamma && ln_beta && y && mean && rstd, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(A) && CCD_ALIGNED16(B) && CCD_ALIGNED16(C) && CCD_ALIGNED16(resid) && CCD_ALIGNED16(y), CCD_EINVAL);
    if (M == 0) return CCD_OK;
    CCD_CHECK(M > 0 && N > 0 && K > 0 && rows_per_sample > 0, CCD_EINVAL);
    CCD_CHECK(N <= 512 && N % 8 == 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && ldr % 4 == 0 &&
              ldy % 8 == 0, CCD_ESHAPE);
    CCD_CHECK(((long)M * lda + K) * 2 < CCD_MAX_OPERAND_BYTES && ((long)N * ldb + K) * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    const int cus = ccd_grid_cus((M + ccd::RG_BM - 1) / ccd::RG_BM);
    // (measured at N = 384: 0.131 ms with gemm_row384.h's 8-wave tile - whose residual rows stream in under the staging
    // barrier - against 0.198 ms here, where the epilogue starts after the last MFMA: the row-owner kernel is the default
    // only where the other does not exist, N = 512; policy rowgemm = 2 forces it)
    if (ccd_policy().rowgemm && (N == 512 || ccd_policy().rowgemm == 2) && (N == 128 || N == 256 || N == 384 || N == 512) &&
        K % (64 * ccd::rg_ring(N)) == 0 &&
        ((long)M + (long)cus * ccd::RG_BM) * lda * 2 < CCD_MAX_OPERAND_BYTES && (long)M * ldr * 4 < CCD_MAX_OPERAND_BYTES &&
        (long)M * ldc * 4 < CCD_MAX_OPERAND_BYTES) {
        ccd::RowGemmParams q = ccd::RowGemmParams();
        q.A = A; q.lda = lda; q.W = B; q.ldw = ldb; q.M = M; q.K = K; q.gamma = ln_gamma; q.ln_beta = ln_beta; q.ln_eps = ln_eps;
        q.bias = bias; q.resid = resid; q.ldr = ldr; q.out = C; q.ldc = ldc; q.ln_y = y; q.ld_y = ldy; q.ln_mean = mean;
        q.ln_rstd = rstd; q.rowscale = rowscale; q.rows_per_sample = rows_per_sample; q.lab = ccd_policy().lab;
        const int tiles = (M + ccd::RG_BM - 1) / ccd::RG_BM, smem = ccd::rg_smem_bytes(N);
        const dim3 grid(tiles < cus ? tiles : cus), block(ccd::RG_THREADS);
        if (N == 512) CCD_LAUNCH((ccd::rowgemm_kernel<512, ccd::rg_ring(512), ccd::RG_RESID_LN>), grid, block, smem, stream, q);
        else if (N == 384 && ccd_policy().rowgemm_adma)
            CCD_LAUNCH((ccd::rowgemm_kernel<384, 3, ccd::RG_RESID_LN, true>), grid, block, ccd::rg_smem_bytes_adma(384), stream, q);
        else if (N == 384) CCD_LAUNCH((ccd::rowgemm_kernel<384, ccd::rg_ring(384), ccd::RG_RESID_LN>), grid, block, smem, stream, q);
        else if (N == 256) CCD_LAUNCH((ccd::rowgemm_kernel<256, ccd::rg_ring(256), ccd::RG_RESID_LN>), grid, block, smem, stream, q);
        else CCD_LAUNCH((ccd::rowgemm_kernel<128, ccd::rg_ring(128), ccd::RG_RESID_LN>), grid, block, smem, stream, q);
        return ccd_rt_last_error();
    }
    CCD_CHECK(N <= ccd::GR_BN, CCD_ESHAPE);
    ccd::GemmParams p = ccd::GemmParams();
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.C = C; p.ldc = ldc; p.bias = bias;
    p.resid = resid; p.ldr = ldr; p.rowscale = rowscale; p.rows_per_sample = rows_per_sample; p.alpha = 1.0f; p.rows_mul = 1;
    p.k_per_split = K;
    p.rps_shift = -1;
    for (int sft = 0; sft < 31; ++sft) if ((1 << sft) == rows_per_sample) p.rps_shift = sft;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_eps = ln_eps; p.ln_y = y; p.ld_y = ldy; p.ln_mean = mean; p.ln_rstd = rstd;
    return ccd_launch_gemm_row384(p, 7 /* EPI_RESID_LN */, stream, cus);
}

struct CcdLnTap {       // a second LayerNorm backward of the same rows (rowgemm.h: TAP), or tap_dy == nullptr
    const ccd_bf16* tap_dy = nullptr;
    long ld_tap = 0;
    const float* tap_gamma = nullptr;
    float* tap_dgamma = nullptr;
    float* tap_dbeta = nullptr;
};
static int ccd_gemm_nt_lnbwd_any(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, const float* x, long ldx,
                                 const float* mean, const float* rstd, const float* gamma, void* g, int g16, long ldg, int accumulate,
                                 float* dgamma, float* dbeta, ccd_bf16* gb, long ldgb, const float* rowscale, int rows_per_sample,
                                 float* dbias, void* stream, const CcdLnTap& tap = CcdLnTap()) {
    if (tap.tap_dy) {       // the tap rides on the row-owner kernel of the bf16 gradient stream at N = 384 (the ViT-Small path) only
        CCD_CHECK(tap.tap_gamma && tap.tap_dgamma && tap.tap_dbeta && CCD_ALIGNED16(tap.tap_dy) && tap.ld_tap % 8 == 0, CCD_EINVAL);
        CCD_CHECK(g16 && N == 384 && ccd_policy().rowgemm && K % (64 * ccd::rg_ring(N)) == 0 && (long)M * tap.ld_tap * 2 < CCD_MAX_OPERAND_BYTES,
                  CCD_ESHAPE);
    }
    CCD_CHECK(A && B && x && mean && rstd && gamma && g && dgamma && dbeta, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(A) && CCD_ALIGNED16(B) && CCD_ALIGNED16(x) && CCD_ALIGNED16(g) && CCD_ALIGNED16(gb), CCD_EINVAL);
    if (M == 0) return CCD_OK;
    CCD_CHECK(M > 0 && N > 0 && K > 0 && (!rowscale || rows_per_sample > 0), CCD_EINVAL);
    CCD_CHECK(N <= 512 && N % 8 == 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldx % 4 == 0 && ldg % 4 == 0 &&
              (!gb || ldgb % 4 == 0), CCD_ESHAPE);
    CCD_CHECK(((long)M * lda + K) * 2 < CCD_MAX_OPERAND_BYTES && ((long)N * ldb + K) * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    const int cus = ccd_grid_cus((M + ccd::RG_BM - 1) / ccd::RG_BM);
    if (ccd_policy().rowgemm && (N == 128 || N == 256 || N == 384 || N == 512) && K % (64 * ccd::rg_ring(N)) == 0 && ldg % 4 == 0 &&
        ((long)M + (long)cus * ccd::RG_BM) * lda * 2 < CCD_MAX_OPERAND_BYTES && (long)M * ldx * 4 < CCD_MAX_OPERAND_BYTES &&
        (long)M * ldg * 4 < CCD_MAX_OPERAND_BYTES && (!g16 || (N <= 384 && ldg % 8 == 0))) {
        ccd::RowGemmParams q;
        q.A = A; q.lda = lda; q.W = B; q.ldw = ldb; q.M = M; q.K = K; q.x = x; q.ldx = ldx; q.mean = mean; q.rstd = rstd;
        q.gamma = gamma; q.g = g; q.ldg = ldg; q.accumulate = accumulate; q.dgamma = dgamma; q.dbeta = dbeta; q.gb = gb;
        q.ld_gb = ldgb; q.rowscale = gb ? rowscale : nullptr; q.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
        q.dbias = gb ? dbias : nullptr; q.lab = ccd_policy().lab;
        q.bias = nullptr; q.resid = nullptr; q.out = nullptr; q.ln_beta = nullptr; q.ln_y = nullptr; q.ln_mean = q.ln_rstd = nullptr;
        q.ldr = q.ldc = q.ld_y = 0; q.ln_eps = 0.f;
        q.tap_dy = tap.tap_dy; q.ld_tap = tap.ld_tap; q.tap_gamma = tap.tap_gamma; q.tap_dgamma = tap.tap_dgamma; q.tap_dbeta = tap.tap_dbeta;
        const int tiles = (M + ccd::RG_BM - 1) / ccd::RG_BM, smem = ccd::rg_smem_bytes(N);
        const dim3 grid(tiles < cus ? tiles : cus), block(ccd::RG_THREADS);
        if (g16 && tap.tap_dy) {
            if (ccd_policy().rowgemm_adma)
                CCD_LAUNCH((ccd::rowgemm_kernel<384, 3, ccd::RG_LNBWD, true, true, true>), grid, block, ccd::rg_smem_bytes_adma(384, true), stream, q);
            else CCD_LAUNCH((ccd::rowgemm_kernel<384, ccd::rg_ring(384), ccd::RG_LNBWD, false, true, true>), grid, block, ccd::rg_smem_bytes(384, true), stream, q);
            return ccd_rt_last_error();
        }
        if (g16) {             // round 6: the residual-gradient stream in bf16
            if (N == 384 && ccd_policy().rowgemm_adma)
                CCD_LAUNCH((ccd::rowgemm_kernel<384, 3, ccd::RG_LNBWD, true, true>), grid, block, ccd::rg_smem_bytes_adma(384), stream, q);
            else if (N == 384) CCD_LAUNCH((ccd::rowgemm_kernel<384, ccd::rg_ring(384), ccd::RG_LNBWD, false, true>), grid, block, smem, stream, q);
            else if (N == 256) CCD_LAUNCH((ccd::rowgemm_kernel<256, ccd::rg_ring(256), ccd::RG_LNBWD, false, true>), grid, block, smem, stream, q);
            else CCD_LAUNCH((ccd::rowgemm_kernel<128, ccd::rg_ring(128), ccd::RG_LNBWD, false, true>), grid, block, smem, stream, q);
            return ccd_rt_last_error();
        }
        if (N == 512) CCD_LAUNCH((ccd::rowgemm_kernel<512, ccd::rg_ring(512), ccd::RG_LNBWD>), grid, block, smem, stream, q);
        else if (N == 384 && ccd_policy().rowgemm_adma)   // round 4: the activation rows by LDS-DMA too
            CCD_LAUNCH((ccd::rowgemm_kernel<384, 3, ccd::RG_LNBWD, true>), grid, block, ccd::rg_smem_bytes_adma(384), stream, q);
        else if (N == 384) CCD_LAUNCH((ccd::rowgemm_kernel<384, ccd::rg_ring(384), ccd::RG_LNBWD>), grid, block, smem, stream, q);
        else if (N == 256) CCD_LAUNCH((ccd::rowgemm_kernel<256, ccd::rg_ring(256), ccd::RG_LNBWD>), grid, block, smem, stream, q);
        else CCD_LAUNCH((ccd::rowgemm_kernel<128, ccd::rg_ring(128), ccd::RG_LNBWD>), grid, block, smem, stream, q);
        return ccd_rt_last_error();
    }
    CCD_CHECK(N <= ccd::GR_BN && !g16, CCD_ESHAPE);       // (the bf16 stream exists in the row-owner kernels only)
    ccd::GemmParams p = ccd::GemmParams();
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.C = g; p.ldc = ldg; p.resid = x; p.ldr = ldx;
    p.rowscale = gb ? rowscale : nullptr; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; p.alpha = 1.0f;
    p.rows_mul = 1; p.k_per_split = K;
    p.rps_shift = -1;
    for (int sft = 0; sft < 31; ++sft) if ((1 << sft) == p.rows_per_sample) p.rps_shift = sft;
    p.ln_gamma = gamma; p.ln_mean = const_cast<float*>(mean); p.ln_rstd = const_cast<float*>(rstd);
    p.lnb_accumulate = accumulate; p.lnb_gb = gb; p.ld_gb = ldgb; p.lnb_dgamma = dgamma; p.lnb_dbeta = dbeta; p.lnb_dbias = dbias;
    return ccd_launch_gemm_row384(p, 8 /* EPI_LNBWD */, stream, cus);
}
int ccd_gemm_nt_lnbwd(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, const float* x, long ldx,
                      const float* mean, const float* rstd, const float* gamma, float* g, long ldg, int accumulate,
                      float* dgamma, float* dbeta, ccd_bf16* gb, long ldgb, const float* rowscale, int rows_per_sample,
                      float* dbias, void* stream) {
    return ccd_gemm_nt_lnbwd_any(A, lda, B, ldb, M, N, K, x, ldx, mean, rstd, gamma, g, 0, ldg, accumulate, dgamma, dbeta, gb, ldgb, rowscale,
                                 rows_per_sample, dbias, stream);
}
int ccd_gemm_nt_lnbwd_g16(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, const float* x, long ldx,
                          const float* mean, const float* rstd, const float* gamma, ccd_bf16* g, long ldg, int accumulate,
                          float* dgamma, float* dbeta, ccd_bf16* gb, long ldgb, const float* rowscale, int rows_per_sample,
                          float* dbias, void* stream) {
    return ccd_gemm_nt_lnbwd_any(A, lda, B, ldb, M, N, K, x, ldx, mean, rstd, gamma, g, 1, ldg, accumulate, dgamma, dbeta, gb, ldgb, rowscale,
                                 rows_per_sample, dbias, stream);
}

int ccd_gemm_nt_lnbwd_tap_g16(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, const float* x, long ldx,
                              const float* mean, const float* rstd, const float* gamma, ccd_bf16* g, long ldg, int accumulate,
                              float* dgamma, float* dbeta, ccd_bf16* gb, long ldgb, const float* rowscale, int rows_per_sample,
                              float* dbias, const ccd_bf16* tap_dy, long ld_tap, const float* tap_gamma, float* tap_dgamma,
                              float* tap_dbeta, void* stream) {
    CCD_CHECK(tap_dy, CCD_EINVAL);
    CcdLnTap tap;
    tap.tap_dy = tap_dy; tap.ld_tap = ld_tap; tap.tap_gamma = tap_gamma; tap.tap_dgamma = tap_dgamma; tap.tap_dbeta = tap_dbeta;
    return ccd_gemm_nt_lnbwd_any(A, lda, B, ldb, M, N, K, x, ldx, mean, rstd, gamma, g, 1, ldg, accumulate, dgamma, dbeta, gb, ldgb, rowscale,
                                 rows_per_sample, dbias, stream, tap);
}

int ccd_mlp_fused(const ccd_bf16* y, long ldy, const ccd_bf16* w1, long ld1, const float* b1, const ccd_bf16* w2, long ld2,
                  const float* b2, const float* resid, long ldr, const float* rowscale, int rows_per_sample, float* out,
                  long ldc, const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* ln_y, long ld_y,
                  float* ln_mean, float* ln_rstd, ccd_bf16* u, long ldu, ccd_bf16* gact, long ldga, int M, int E, int H, void* stream) {
    CCD_CHECK(y && w1 && b1 && w2 && b2 && resid && out && ln_gamma && ln_beta && ln_y && ln_mean && ln_rstd, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(y) && CCD_ALIGNED16(w1) && CCD_ALIGNED16(w2) && CCD_ALIGNED16(resid) && CCD_ALIGNED16(out) &&
              CCD_ALIGNED16(ln_y) && CCD_ALIGNED16(u), CCD_EINVAL);
    if (M == 0) return CCD_OK;
    CCD_CHECK(M > 0 && H > 0 && (!rowscale || rows_per_sample > 0), CCD_EINVAL);
    CCD_CHECK((E == 128 || E == 256 || E == 384 || E == 512) && H % 64 == 0 && ldy % 8 == 0 && ld1 % 8 == 0 && ld2 % 8 == 0 && ldr % 4 == 0 &&
              ldc % 4 == 0 && ld_y % 8 == 0 && (!u || ldu % 8 == 0) && (!gact || (u && ldga % 8 == 0 && CCD_ALIGNED16(gact))), CCD_ESHAPE);
    CCD_CHECK((long)H * ld1 * 2 < CCD_MAX_OPERAND_BYTES && (long)E * ld2 * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    const int smem = ccd::mlp_smem_bytes(E, H);
    CCD_CHECK(smem <= 160 * 1024, CCD_ESHAPE);
    ccd::MlpParams p;
    p.y = y; p.ldy_in = ldy; p.w1 = w1; p.ld1 = ld1; p.b1 = b1; p.w2 = w2; p.ld2 = ld2; p.b2 = b2; p.resid = resid; p.ldr = ldr;
    p.rowscale = rowscale; p.rows_per_sample = rowscale ? rows_per_sample : 1; p.out = out; p.ldc = ldc;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_eps = ln_eps; p.ln_y = ln_y; p.ld_y = ld_y; p.ln_mean = ln_mean;
    p.ln_rstd = ln_rstd; p.u = u; p.ldu = ldu; p.gact = gact; p.ldga = ldga; p.M = M; p.H = H; p.lab = ccd_policy().lab;
    p.a = nullptr; p.lda = 0; p.wp = nullptr; p.ldp = 0; p.bp = nullptr; p.rowscale1 = nullptr; p.ln2_gamma = p.ln2_beta = nullptr;
    p.xmid = nullptr; p.ldxm = 0; p.y2 = nullptr; p.ldy2 = 0; p.mean2 = p.rstd2 = nullptr;
    p.tap_gamma = p.tap_beta = nullptr; p.tap_y = nullptr; p.ld_tap = 0;
    const int tiles = (M + ccd::MLP_BM - 1) / ccd::MLP_BM, cus = ccd_grid_cus(tiles);
    const dim3 grid(tiles < cus ? tiles : cus), block(ccd::MLP_THREADS);
    if (E == 512) {
        if (u) CCD_LAUNCH((ccd::mlp_fused_kernel<512, true>), grid, block, smem, stream, p);
        else CCD_LAUNCH((ccd::mlp_fused_kernel<512, false>), grid, block, smem, stream, p);
    } else if (E == 384) {
        if (u) CCD_LAUNCH((ccd::mlp_fused_kernel<384, true>), grid, block, smem, stream, p);
        else CCD_LAUNCH((ccd::mlp_fused_kernel<384, false>), grid, block, smem, stream, p);
    } else if (E == 256) {
        if (u) CCD_LAUNCH((ccd::mlp_fused_kernel<256, true>), grid, block, smem, stream, p);
        else CCD_LAUNCH((ccd::mlp_fused_kernel<256, false>), grid, block, smem, stream, p);
    } else {
        if (u) CCD_LAUNCH((ccd::mlp_fused_kernel<128, true>), grid, block, smem, stream, p);
        else CCD_LAUNCH((ccd::mlp_fused_kernel<128, false>), grid, block, smem, stream, p);
    }
    return ccd_rt_last_error();
}

static int ccd_proj_mlp_fused_impl(const ccd_bf16* a, long lda, const ccd_bf16* wp, long ldp, const float* bp, const float* resid, long ldr,
                       const float* rowscale1, const float* ln2_gamma, const float* ln2_beta, float* xmid, long ldxm, ccd_bf16* y2,
                       long ldy2, float* mean2, float* rstd2, const ccd_bf16* w1, long ld1, const float* b1, const ccd_bf16* w2,
                       long ld2, const float* b2, const float* rowscale2, int rows_per_sample, float* out, long ldc,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* ln_y, long ld_y, float* ln_mean,
                       float* ln_rstd, ccd_bf16* u, long ldu, ccd_bf16* gact, long ldga, const float* tap_gamma, const float* tap_beta,
                       ccd_bf16* tap_y, long ld_tap, int M, int E, int H, void* stream) {
    CCD_CHECK(!gact || (u && ldga % 8 == 0 && CCD_ALIGNED16(gact)), CCD_EINVAL);
    CCD_CHECK((tap_y != nullptr) == (tap_gamma != nullptr) && (tap_y != nullptr) == (tap_beta != nullptr) && CCD_ALIGNED16(tap_y) &&
              (!tap_y || ld_tap % 8 == 0), CCD_EINVAL);
    CCD_CHECK(a && wp && bp && resid && ln2_gamma && ln2_beta && w1 && b1 && w2 && b2 && out && ln_gamma && ln_beta && ln_y && ln_mean &&
              ln_rstd, CCD_EINVAL);
    CCD_CHECK((xmid != nullptr) == (y2 != nullptr) && (xmid != nullptr) == (mean2 != nullptr) && (xmid != nullptr) == (rstd2 != nullptr) &&
              (xmid != nullptr) == (u != nullptr), CCD_EINVAL);       // what the backward pass reads: all five or none
    CCD_CHECK(!rowscale2 || xmid, CCD_EINVAL);    // a dropped MLP branch reads x_mid back (only a pass that trains drops branches)
    CCD_CHECK(CCD_ALIGNED16(a) && CCD_ALIGNED16(wp) && CCD_ALIGNED16(w1) && CCD_ALIGNED16(w2) && CCD_ALIGNED16(resid) && CCD_ALIGNED16(out) &&
              CCD_ALIGNED16(ln_y) && CCD_ALIGNED16(u) && CCD_ALIGNED16(xmid) && CCD_ALIGNED16(y2), CCD_EINVAL);
    if (M == 0) return CCD_OK;
    CCD_CHECK(M > 0 && H > 0 && ((!rowscale1 && !rowscale2) || rows_per_sample > 0), CCD_EINVAL);
    // a DropPath scale is read once per 128-row tile (a dropped branch's weight pieces are skipped): tiles must not span samples
    CCD_CHECK((!rowscale1 && !rowscale2) || rows_per_sample % ccd::MLP_BM == 0, CCD_ESHAPE);
    CCD_CHECK((E == 128 || E == 256 || E == 384) && H % 64 == 0 && lda % 8 == 0 && ldp % 8 == 0 && ld1 % 8 == 0 && ld2 % 8 == 0 && ldr % 4 == 0 &&
              ldc % 4 == 0 && ld_y % 8 == 0 && (!u || ldu % 8 == 0) && (!xmid || (ldxm % 4 == 0 && ldy2 % 8 == 0)), CCD_ESHAPE);
    CCD_CHECK((long)H * ld1 * 2 < CCD_MAX_OPERAND_BYTES && (long)E * ld2 * 2 < CCD_MAX_OPERAND_BYTES && (long)E * ldp * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    const int smem = ccd::mlp_smem_bytes(E, H, true);
    CCD_CHECK(smem <= 160 * 1024, CCD_ESHAPE);
    ccd::MlpParams p;
    p.y = nullptr; p.ldy_in = 0; p.w1 = w1; p.ld1 = ld1; p.b1 = b1; p.w2 = w2; p.ld2 = ld2; p.b2 = b2; p.resid = resid; p.ldr = ldr;
    p.rowscale = rowscale2; p.rows_per_sample = (rowscale1 || rowscale2) ? rows_per_sample : ccd::MLP_BM; p.out = out; p.ldc = ldc;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_eps = ln_eps; p.ln_y = ln_y; p.ld_y = ld_y; p.ln_mean = ln_mean;
    p.ln_rstd = ln_rstd; p.u = u; p.ldu = ldu; p.gact = gact; p.ldga = ldga; p.M = M; p.H = H; p.lab = ccd_policy().lab;
    p.a = a; p.lda = lda; p.wp = wp; p.ldp = ldp; p.bp = bp; p.rowscale1 = rowscale1; p.ln2_gamma = ln2_gamma; p.ln2_beta = ln2_beta;
    p.xmid = xmid; p.ldxm = ldxm; p.y2 = y2; p.ldy2 = ldy2; p.mean2 = mean2; p.rstd2 = rstd2;
    p.tap_gamma = tap_gamma; p.tap_beta = tap_beta; p.tap_y = tap_y; p.ld_tap = ld_tap;
    const int tiles = (M + ccd::MLP_BM - 1) / ccd::MLP_BM, cus = ccd_grid_cus(tiles);
    const dim3 grid(tiles < cus ? tiles : cus), block(ccd::MLP_THREADS);
    if (E == 384) {
        if (u) CCD_LAUNCH((ccd::mlp_fused_kernel<384, true, true>), grid, block, smem, stream, p);
        else CCD_LAUNCH((ccd::mlp_fused_kernel<384, false, true>), grid, block, smem, stream, p);
    } else if (E == 256) {
        if (u) CCD_LAUNCH((ccd::mlp_fused_kernel<256, true, true>), grid, block, smem, stream, p);
        else CCD_LAUNCH((ccd::mlp_fused_kernel<256, false, true>), grid, block, smem, stream, p);
    } else {
        if (u) CCD_LAUNCH((ccd::mlp_fused_kernel<128, true, true>), grid, block, smem, stream, p);
        else CCD_LAUNCH((ccd::mlp_fused_kernel<128, false, true>), grid, block, smem, stream, p);
    }
    return ccd_rt_last_error();
}
int ccd_proj_mlp_fused(const ccd_bf16* a, long lda, const ccd_bf16* wp, long ldp, const float* bp, const float* resid, long ldr,
                       const float* rowscale1, const float* ln2_gamma, const float* ln2_beta, float* xmid, long ldxm, ccd_bf16* y2,
                       long ldy2, float* mean2, float* rstd2, const ccd_bf16* w1, long ld1, const float* b1, const ccd_bf16* w2,
                       long ld2, const float* b2, const float* rowscale2, int rows_per_sample, float* out, long ldc,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* ln_y, long ld_y, float* ln_mean,
                       float* ln_rstd, ccd_bf16* u, long ldu, const float* tap_gamma, const float* tap_beta, ccd_bf16* tap_y, long ld_tap,
                       int M, int E, int H, void* stream) {
    return ccd_proj_mlp_fused_impl(a, lda, wp, ldp, bp, resid, ldr, rowscale1, ln2_gamma, ln2_beta, xmid, ldxm, y2, ldy2, mean2, rstd2, w1, ld1,
                                   b1, w2, ld2, b2, rowscale2, rows_per_sample, out, ldc, ln_gamma, ln_beta, ln_eps, ln_y, ld_y, ln_mean, ln_rstd,
                                   u, ldu, nullptr, 0, tap_gamma, tap_beta, tap_y, ld_tap, M, E, H, stream);
}
int ccd_proj_mlp_fused_gact(const ccd_bf16* a, long lda, const ccd_bf16* wp, long ldp, const float* bp, const float* resid, long ldr,
                            const float* rowscale1, const float* ln2_gamma, const float* ln2_beta, float* xmid, long ldxm, ccd_bf16* y2,
                            long ldy2, float* mean2, float* rstd2, const ccd_bf16* w1, long ld1, const float* b1, const ccd_bf16* w2,
                            long ld2, const float* b2, const float* rowscale2, int rows_per_sample, float* out, long ldc,
                            const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* ln_y, long ld_y, float* ln_mean,
                            float* ln_rstd, ccd_bf16* u, long ldu, ccd_bf16* gact, long ldga, const float* tap_gamma, const float* tap_beta,
                            ccd_bf16* tap_y, long ld_tap, int M, int E, int H, void* stream) {
    CCD_CHECK(gact, CCD_EINVAL);
    return ccd_proj_mlp_fused_impl(a, lda, wp, ldp, bp, resid, ldr, rowscale1, ln2_gamma, ln2_beta, xmid, ldxm, y2, ldy2, mean2, rstd2, w1, ld1,
                                   b1, w2, ld2, b2, rowscale2, rows_per_sample, out, ldc, ln_gamma, ln_beta, ln_eps, ln_y, ld_y, ln_mean, ln_rstd,
                                   u, ldu, gact, ldga, tap_gamma, tap_beta, tap_y, ld_tap, M, E, H, stream);
}

int ccd_mlp_bwd_fused(const ccd_bf16* gb, long ldgb, const ccd_bf16* w2t, long ld2, const ccd_bf16* w1t, long ld1, const ccd_bf16* u,
                      long ldu, ccd_bf16* du, long lddu, float* db1, const float* x, long ldx, const float* mean, const float* rstd,
                      const float* gamma, ccd_bf16* g, long ldg, int accumulate, float* dgamma, float* dbeta, ccd_bf16* gb_out, long ld_gbo,
                      const float* rowscale, int rows_per_sample, float* dbias, int M, int E, int H, void* stream) {
    CCD_CHECK(gb && w2t && w1t && u && du && db1 && x && mean && rstd && gamma && g && dgamma && dbeta, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(gb) && CCD_ALIGNED16(w2t) && CCD_ALIGNED16(w1t) && CCD_ALIGNED16(u) && CCD_ALIGNED16(du) && CCD_ALIGNED16(x) &&
              CCD_ALIGNED16(g) && CCD_ALIGNED16(gb_out), CCD_EINVAL);
    CCD_CHECK(gb_out != gb, CCD_EINVAL);                    // the weight-gradient launch that follows still reads gb
    if (M == 0) return CCD_OK;
    CCD_CHECK(M > 0 && H > 0 && (!rowscale || rows_per_sample > 0), CCD_EINVAL);
    // (H % 128: the two u images of a wave are used in turn by the chunks of 64 hidden units, across tiles)
    CCD_CHECK((E == 256 || E == 384) && H % 128 == 0 && ldgb % 8 == 0 && ld2 % 8 == 0 && ld1 % 8 == 0 && ldu % 8 == 0 &&
              lddu % 8 == 0 && ldx % 4 == 0 && ldg % 8 == 0 && (!gb_out || ld_gbo % 8 == 0), CCD_ESHAPE);
    CCD_CHECK((long)H * ld2 * 2 < CCD_MAX_OPERAND_BYTES && (long)E * ld1 * 2 < CCD_MAX_OPERAND_BYTES &&
              ((long)M + 128) * lddu * 2 < CCD_MAX_OPERAND_BYTES && ((long)M + 128) * ldx * 4 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    const int smem = ccd::mb_smem_bytes(E, H);
    CCD_CHECK(smem <= 160 * 1024, CCD_ESHAPE);
    ccd::MlpBwdParams p;
    p.gb = gb; p.ld_gb_in = ldgb; p.w2t = w2t; p.ld2 = ld2; p.w1t = w1t; p.ld1 = ld1; p.u = u; p.ldu = ldu; p.du = du; p.lddu = lddu;
    p.db1 = db1; p.x = x; p.ldx = ldx; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.g = g; p.ldg = ldg; p.accumulate = accumulate;
    p.dgamma = dgamma; p.dbeta = dbeta; p.gb_out = gb_out; p.ld_gbo = ld_gbo; p.rowscale = gb_out ? rowscale : nullptr;
    p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; p.dbias = gb_out ? dbias : nullptr; p.M = M; p.H = H;
    const int tiles = (M + ccd::MB_BM - 1) / ccd::MB_BM, cus = ccd_grid_cus(tiles);
    const dim3 grid(tiles < cus ? tiles : cus), block(ccd::MB_THREADS);
    if (E == 384) CCD_LAUNCH((ccd::mlp_bwd_fused_kernel<384>), grid, block, smem, stream, p);
    else CCD_LAUNCH((ccd::mlp_bwd_fused_kernel<256>), grid, block, smem, stream, p);
    return ccd_rt_last_error();
}


static int ccd_gemm_tn_impl(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, int Mc, int epilogue, float* C,
                            long ldc, float alpha, int splits, const int* d_rows, int rows_mul, float* colsum_a, void* stream);
// gemm_tn384.h: one group of (P / TP) (Q / TQ) [+ the second problem's] workgroups per contraction slice, whole groups per XCD
// (workgroup b runs on XCD b % 8).  CCD_ESHAPE: the tiles of one slice do not fit the grid - the caller takes another kernel.
// workgroup geometry for a P x Q product: 0 = 384 x 192 tiles, 2 = 512 x 128 (the E = 512 shapes, policy gemm_tn384_geom = 2);
// -1 = neither divides the shape
static int ccd_tn384_geom(int P, int Q, int Mc) {
    if (Mc % ccd::TN3_BK != 0 || Mc < 2048) return -1;
    if (P % 384 == 0 && Q % 192 == 0) return 0;
    if (ccd_policy().gemm_tn384_geom == 2 && P % 512 == 0 && Q % 128 == 0) return 2;    // (measured on vit_base: 44.6 vs 44.3 ms - opt-in)
    return -1;
}
static int ccd_tn384_tiles(int geom, int P, int Q) {
    return geom == 0 ? (P / 384) * (Q / 192) : (P / 512) * (Q / 128);
}
static int ccd_launch_tn384(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, float* C, long ldc,
                            const ccd_bf16* A2, long lda2, const ccd_bf16* B2, long ldb2, int P2, int Q2, float* C2, long ldc2,
                            int Mc, float alpha, float* lab_out, float* ws, long ws_floats, void* stream) {
    ccd::GemmParams p = ccd::GemmParams();
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.M = P; p.N = Q; p.K = Mc; p.C = C; p.ldc = ldc;
    p.A2 = A2; p.B2 = B2; p.lda2 = lda2; p.ldb2 = ldb2; p.M2 = P2; p.N2 = Q2; p.C2 = C2; p.ldc2 = ldc2;
    p.alpha = alpha; p.rps_shift = ccd_policy().lab; p.colsum_a = lab_out;
    const int geom = ccd_tn384_geom(P, Q, Mc);
    if (geom < 0 || (P2 > 0 && ccd_tn384_geom(P2, Q2, Mc) != geom)) return CCD_ESHAPE;
    if (geom == 2) return ccd_launch_tn384_geom<4, 2, 3, 4, 2>(p, Mc, ws, ws_floats, stream);
    return ccd_launch_tn384_geom<4, 2, 4, 3, 3>(p, Mc, ws, ws_floats, stream);
}
int ccd_gemm_tn_pair(const ccd_bf16* A1, long lda1, const ccd_bf16* B1, long ldb1, int P1, int Q1, float* C1, long ldc1,
                     const ccd_bf16* A2, long lda2, const ccd_bf16* B2, long ldb2, int P2, int Q2, float* C2, long ldc2, int Mc,
                     void* stream) {
    return ccd_gemm_tn_pair_ws(A1, lda1, B1, ldb1, P1, Q1, C1, ldc1, A2, lda2, B2, ldb2, P2, Q2, C2, ldc2, Mc, nullptr, 0, stream);
}
long ccd_gemm_tn_pair_ws_floats(int P1, int Q1, int P2, int Q2) {
    // one plane per contraction slice and problem; a launch never has more slices than workgroup slots (geometry 0: one 8-wave
    // workgroup per CU; geometry 2 - 512 x 128 tiles, policy gemm_tn384_geom = 2 - the same).  Whatever geometry ccd_tn384_geom picks
    // for the pair (both problems the same one), or 0 where the grouped kernel does not apply.
    const int g1 = ccd_tn384_geom(P1, Q1, 2048), g2 = ccd_tn384_geom(P2, Q2, 2048);
    if (g1 < 0 || g1 != g2) return 0;
    const long t1 = ccd_tn384_tiles(g1, P1, Q1), t2 = ccd_tn384_tiles(g2, P2, Q2);
    if (t1 < 1 || t2 < 1) return 0;
    const long cus = ccd_rt_num_cus();
    return (cus / t1 + 8) * (long)P1 * Q1 + (cus / t2 + 8) * (long)P2 * Q2;
}
int ccd_gemm_tn_pair_ws(const ccd_bf16* A1, long lda1, const ccd_bf16* B1, long ldb1, int P1, int Q1, float* C1, long ldc1,
                        const ccd_bf16* A2, long lda2, const ccd_bf16* B2, long ldb2, int P2, int Q2, float* C2, long ldc2, int Mc,
                        float* ws, long ws_floats, void* stream) {
    CCD_CHECK(A1 && B1 && C1 && A2 && B2 && C2, CCD_EINVAL);
    CCD_CHECK(!ws || (CCD_ALIGNED16(ws) && ws_floats >= 0), CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(A1) && CCD_ALIGNED16(B1) && CCD_ALIGNED16(C1) && CCD_ALIGNED16(A2) && CCD_ALIGNED16(B2) && CCD_ALIGNED16(C2),
              CCD_EINVAL);
    CCD_CHECK(P1 > 0 && Q1 > 0 && P2 > 0 && Q2 > 0 && Mc >= 0, CCD_EINVAL);
    if (Mc == 0) return CCD_OK;
    CCD_CHECK(lda1 % 8 == 0 && ldb1 % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && ldc1 % 4 == 0 && ldc2 % 4 == 0, CCD_ESHAPE);
    if (ccd_policy().gemm_tn384 && ccd_policy().gemm_tn384 != 2 && ccd_tn384_geom(P1, Q1, Mc) >= 0 &&
        ccd_tn384_geom(P1, Q1, Mc) == ccd_tn384_geom(P2, Q2, Mc)) {
        const int rc = ccd_launch_tn384(A1, lda1, B1, ldb1, P1, Q1, C1, ldc1, A2, lda2, B2, ldb2, P2, Q2, C2, ldc2, Mc, 1.0f, nullptr,
                                        ccd_policy().tn_ws ? ws : nullptr, ws_floats, stream);
        if (rc != CCD_ESHAPE) return rc;
    }
    const int rc = ccd_gemm_tn_impl(A1, lda1, B1, ldb1, P1, Q1, Mc, CCD_EPI_ATOMIC, C1, ldc1, 1.0f, 0, nullptr, 1, nullptr, stream);
    if (rc != CCD_OK) return rc;
    return ccd_gemm_tn_impl(A2, lda2, B2, ldb2, P2, Q2, Mc, CCD_EPI_ATOMIC, C2, ldc2, 1.0f, 0, nullptr, 1, nullptr, stream);
}
int ccd_gemm_tn(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, int Mc, int epilogue, float* C,
                long ldc, float alpha, int splits, const int* d_rows, int rows_mul, void* stream) {
    return ccd_gemm_tn_impl(A, lda, B, ldb, P, Q, Mc, epilogue, C, ldc, alpha, splits, d_rows, rows_mul, nullptr, stream);
}
int ccd_gemm_tn_colsum(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, int Mc, float* C, long ldc,
                       float* colsum_a, int splits, void* stream) {
    CCD_CHECK(colsum_a, CCD_EINVAL);
    return ccd_gemm_tn_impl(A, lda, B, ldb, P, Q, Mc, CCD_EPI_ATOMIC, C, ldc, 1.0f, splits, nullptr, 1, colsum_a, stream);
}
static int ccd_gemm_tn_impl(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, int Mc, int epilogue, float* C,
                            long ldc, float alpha, int splits, const int* d_rows, int rows_mul, float* colsum_a, void* stream) {
    CCD_CHECK(A && B && C, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(A) && CCD_ALIGNED16(B) && CCD_ALIGNED16(C), CCD_EINVAL);
    if (P == 0 || Q == 0 || Mc == 0) return CCD_OK;
    CCD_CHECK(P > 0 && Q > 0 && Mc > 0, CCD_EINVAL);
    CCD_CHECK(P % 8 == 0 && Q % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, CCD_ESHAPE);
    CCD_CHECK(epilogue == CCD_EPI_ATOMIC || epilogue == CCD_EPI_F32, CCD_EINVAL);
    if (ccd_policy().gemm_tn384 && epilogue == CCD_EPI_ATOMIC && !d_rows && (!colsum_a || (ccd_policy().lab & 4)) &&
        ccd_tn384_geom(P, Q, Mc) >= 0 &&
        ccd_tn384_tiles(ccd_tn384_geom(P, Q, Mc), P, Q) >= ccd_policy().gemm_tn384_min_tiles) {
        const int rc = ccd_launch_tn384(A, lda, B, ldb, P, Q, C, ldc, nullptr, 0, nullptr, 0, 0, 0, nullptr, 0, Mc, alpha, colsum_a, nullptr, 0, stream);
        if (rc != CCD_ESHAPE) return rc;
    }
    if (splits < 1) {   // as many slices as fit ONE resident wave of workgroups (2 per CU): one extra workgroup would
                        // run alone after all others and double the kernel time
        const int tiles = ((P + 127) / 128) * ((Q + 127) / 128);
        splits = (2 * ccd_rt_num_cus()) / tiles;
        if (splits < 1) splits = 1;
    }
    int per = (Mc + splits - 1) / splits;
    per = ((per + 63) / 64) * 64;
    splits = (Mc + per - 1) / per;
    CCD_CHECK(epilogue == CCD_EPI_ATOMIC || splits == 1, CCD_EINVAL);
    ccd::GemmParams p = ccd::GemmParams();
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.M = P; p.N = Q; p.K = Mc;
    p.C = C; p.ldc = ldc; p.C2 = nullptr; p.ldc2 = 0; p.bias = nullptr; p.resid = nullptr; p.ldr = 0;
    p.rowscale = nullptr; p.rows_per_sample = 1; p.rps_shift = 0; p.aux = nullptr; p.ldaux = 0;
    p.k_per_split = per; p.m_fastest = 0; p.alpha = alpha; p.d_rows = d_rows; p.rows_mul = rows_mul;
    p.colsum = nullptr; p.colsum_a = colsum_a;
    return ccd_launch_gemm<true>(p, epilogue, splits, stream);
}

// ------------------------------------------------------------------------------------------ LayerNorm
int ccd_ln_fwd(const float* x, const float* gamma, const float* beta, ccd_bf16* y, float* mean, float* rstd, int rows,
               int E, float eps, void* stream) {
    CCD_CHECK(x && gamma && beta && y && mean && rstd, CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && E > 0 && E % 4 == 0 && E <= 1024, CCD_ESHAPE);
    if (ccd::ln_steps(E) == 2)
        CCD_LAUNCH((ccd::ln_fwd_kernel<2>), dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, rows, E, eps);
    else
        CCD_LAUNCH((ccd::ln_fwd_kernel<4>), dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, rows, E, eps);
    return ccd_rt_last_error();
}

static int ccd_ln_bwd_any(const ccd_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma, void* g, int g16,
                          int accumulate, float* dgamma, float* dbeta, ccd_bf16* gb, const float* rowscale, int rows_per_sample,
                          float* dbias, int rows, int E, void* stream) {
    CCD_CHECK(dy && x && mean && rstd && gamma && g && dgamma && dbeta, CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && E > 0 && E % 4 == 0 && E <= 1024, CCD_ESHAPE);
    CCD_CHECK(!rowscale || rows_per_sample > 0, CCD_EINVAL);
    int blocks = ccd_policy().ln_bwd_bpc * ccd_rt_num_cus();   // one resident wave of blocks (measured 187 -> 165 us vs 8 / CU)
    int rpb = (rows + blocks - 1) / blocks;
    rpb = ((rpb + 3) / 4) * 4;
    blocks = (rows + rpb - 1) / rpb;
#define CCD_LN_BWD(ACC, ST, G16) CCD_LAUNCH((ccd::ln_bwd_kernel<ACC, ST, G16>), dim3(blocks), dim3(256), 0, stream, dy, x, mean, rstd, gamma, g, \
                                            dgamma, dbeta, gb, rowscale, rows_per_sample, dbias, rows, E, rpb)
    if (g16) {
        if (ccd::ln_steps(E) == 2) { if (accumulate) CCD_LN_BWD(true, 2, true); else CCD_LN_BWD(false, 2, true); }
        else { if (accumulate) CCD_LN_BWD(true, 4, true); else CCD_LN_BWD(false, 4, true); }
    } else if (ccd::ln_steps(E) == 2) { if (accumulate) CCD_LN_BWD(true, 2, false); else CCD_LN_BWD(false, 2, false); }
    else { if (accumulate) CCD_LN_BWD(true, 4, false); else CCD_LN_BWD(false, 4, false); }
#undef CCD_LN_BWD
    return ccd_rt_last_error();
}
int ccd_ln_bwd(const ccd_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* g,
               int accumulate, float* dgamma, float* dbeta, ccd_bf16* gb, const float* rowscale, int rows_per_sample,
               float* dbias, int rows, int E, void* stream) {
    return ccd_ln_bwd_any(dy, x, mean, rstd, gamma, g, 0, accumulate, dgamma, dbeta, gb, rowscale, rows_per_sample, dbias, rows, E, stream);
}
int ccd_ln_bwd_g16(const ccd_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma, ccd_bf16* g,
                   int accumulate, float* dgamma, float* dbeta, ccd_bf16* gb, const float* rowscale, int rows_per_sample,
                   float* dbias, int rows, int E, void* stream) {
    return ccd_ln_bwd_any(dy, x, mean, rstd, gamma, g, 1, accumulate, dgamma, dbeta, gb, rowscale, rows_per_sample, dbias, rows, E, stream);
}

// ------------------------------------------------------------------------------------------ attention
int ccd_attention_fwd(const ccd_bf16* qkv, ccd_bf16* out, float* lse, int views, int heads, float scale,
                      void* stream) {
    CCD_CHECK(qkv && out && lse, CCD_EINVAL);
    if (views == 0) return CCD_OK;
    CCD_CHECK(views > 0 && heads > 0, CCD_EINVAL);
    // (Round 4 measured the forward with its K / V images written by LDS-DMA and the V^T fragments read by ds_read_b64_tr_b16 - no
    // register transposition, 158 instead of 225 registers: 0.1039 against 0.1027 ms per layer.  With two workgroups per CU one's
    // staging already hides under the other's products; removed again, profiles/r04_attn_onepass_lab.jsonl.)
    CCD_LAUNCH(ccd::attention_fwd_kernel, dim3(views * heads), dim3(256), ccd::ATT_SMEM_BYTES, stream, qkv, out, lse,
               heads, scale);
    return ccd_rt_last_error();
}

long ccd_attention_bwd_ws_floats(int views, int heads) {
    const long nblocks = (long)views * heads, cus = ccd_rt_num_cus();
    return (nblocks < cus ? nblocks : cus) * (long)heads * ccd::ATT_D;
}
// (Measured in round 3 and dropped: the views in 2 / 4 / 8 chunks of launches, so that the dK / dV kernel's re-reads of q, k, v, dO
// would come from the 256-MiB MALL right after the dQ kernel touched them - 0.377 / 0.397 / 0.493 ms per layer against 0.353 in
// one go (profiles/r03_attn_bwd_lab.jsonl): the re-reads are not what bounds the pair, the extra launch tails are pure cost.)
int ccd_attention_bwd(const ccd_bf16* qkv, const ccd_bf16* out, const ccd_bf16* d_out, const float* lse,
                      float* delta_ws, ccd_bf16* d_qkv, int views, int heads, float scale, float* d_qkv_bias, float* bias_ws,
                      const float* dout_colsum_vec, const float* dout_colsum_mat, long ld_mat, void* stream) {
    CCD_CHECK(qkv && out && d_out && lse && delta_ws && d_qkv && (!d_qkv_bias || (bias_ws && dout_colsum_vec)), CCD_EINVAL);
    if (views == 0) return CCD_OK;
    CCD_CHECK(views > 0 && heads > 0 && heads <= ccd::ATTB_MAX_HEADS, CCD_EINVAL);
    const int nblocks = views * heads;                      // persistent: one workgroup per CU walks the (view, head) blocks
    const int cus = ccd_rt_num_cus();
    const int grid = nblocks < cus ? nblocks : cus;
    float* ws = d_qkv_bias ? bias_ws : nullptr;             // [grid][E] partial column sums of dQ
    if (ccd_policy().attn_onepass)          // round 4: one pass, five products (attention_bwd1.h)
        CCD_LAUNCH(ccd::attention_bwd_onepass_kernel, dim3(grid), dim3(512), ccd::ATTB1_SMEM, stream, qkv, out, d_out, lse, d_qkv, ws,
                   heads, scale, nblocks, ccd_policy().lab);
    else {
    CCD_LAUNCH(ccd::attention_bwd_dq_kernel, dim3(grid), dim3(512), ccd::ATTB_DQ_SMEM, stream, qkv,
               out, d_out, lse, delta_ws, d_qkv, ws, heads, scale, nblocks, ccd_policy().attn_skew);
    if (ccd_policy().attn_tr)               // dK / dV on the double-buffered LDS-DMA image with transposing LDS reads
        CCD_LAUNCH(ccd::attention_bwd_dkv_tr_kernel, dim3(grid), dim3(512), ccd::ATTB_DKV_TR_SMEM, stream,
                   qkv, d_out, lse, delta_ws, d_qkv, heads, scale, nblocks, ccd_policy().lab);
    else
        CCD_LAUNCH(ccd::attention_bwd_dkv_kernel, dim3(grid), dim3(512), ccd::ATTB_DKV_SMEM, stream, qkv,
                   d_out, lse, delta_ws, d_qkv, heads, scale, nblocks, ccd_policy().attn_skew);
    }
    if (ws) {
        const int E = heads * ccd::ATT_D;
        CCD_LAUNCH(ccd::qkv_bias_finish_kernel, dim3((E + 63) / 64, 2), dim3(1024), 0, stream, ws, grid, dout_colsum_vec,
                   dout_colsum_mat, ld_mat, E, d_qkv_bias);
    }
    return ccd_rt_last_error();
}

// ------------------------------------------------------------------------------- patch embed & helpers
int ccd_patch_embed_fwd(const float* img, const float* w, const float* bias, const float* pos, float* out, int views,
                        int E, void* stream) {
    CCD_CHECK(img && w && bias && pos && out, CCD_EINVAL);
    if (views == 0) return CCD_OK;
    CCD_CHECK(views > 0 && E > 0, CCD_EINVAL);
    CCD_LAUNCH(ccd::patch_embed_fwd_kernel, dim3(views * ccd::PE_GH), dim3(128), 0, stream, img, w, bias, pos, out, E);
    return ccd_rt_last_error();
}
static int ccd_patch_embed_bwd_any(const float* img, const void* g, int g16, float* d_w, float* d_bias, float* d_pos, ccd_bf16* ws_g,
                                   ccd_bf16* ws_patches, int views, int E, void* stream) {
    if (g16) ws_g = const_cast<ccd_bf16*>(reinterpret_cast<const ccd_bf16*>(g));       // the stream IS the bf16 operand: no copy
    CCD_CHECK(img && g && d_w && d_bias && d_pos && ws_g && ws_patches, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(g) && CCD_ALIGNED16(img) && CCD_ALIGNED16(ws_g) && CCD_ALIGNED16(ws_patches), CCD_EINVAL);
    if (views == 0) return CCD_OK;
    CCD_CHECK(views > 0 && E > 0 && E % 8 == 0, CCD_ESHAPE);
    const int threads = 256 * (E / 4);
    int slices = (4 * ccd_rt_num_cus() * 256 + threads - 1) / threads;       // ~4 workgroups per CU in total
    if (slices > views) slices = views;
    const int vps = (views + slices - 1) / slices;
    slices = (views + vps - 1) / vps;
    if (g16) CCD_LAUNCH(ccd::pos_grad_sum16_kernel, dim3((threads + 255) / 256, slices), dim3(256), 0, stream, (const ccd::bf16_t*)g, d_pos, views, E, vps);
    else CCD_LAUNCH(ccd::pos_grad_cast_kernel, dim3((threads + 255) / 256, slices), dim3(256), 0, stream, (const float*)g, d_pos, ws_g, views,
                    E, vps);
    int rc = ccd_rt_last_error();
    if (rc != CCD_OK) return rc;
    const long tokens = 256L * views;
    CCD_CHECK(tokens < (1L << 31), CCD_ESHAPE);
    rc = ccd_colsum_bf16(ws_g, E, (int)tokens, E, nullptr, 1, d_bias, stream);
    if (rc != CCD_OK) return rc;
    CCD_LAUNCH(ccd::patch_rows_kernel, dim3((unsigned)((tokens * 12 + 255) / 256)), dim3(256), 0, stream, img, ws_patches,
               tokens);
    rc = ccd_rt_last_error();
    if (rc != CCD_OK) return rc;
    return ccd_gemm_tn(ws_g, E, ws_patches, ccd::PE_K, E, ccd::PE_K, (int)tokens, CCD_EPI_ATOMIC, d_w, ccd::PE_K, 1.0f, 0,
                       nullptr, 1, stream);
}
int ccd_patch_embed_bwd(const float* img, const float* g, float* d_w, float* d_bias, float* d_pos, ccd_bf16* ws_g,
                        ccd_bf16* ws_patches, int views, int E, void* stream) {
    return ccd_patch_embed_bwd_any(img, g, 0, d_w, d_bias, d_pos, ws_g, ws_patches, views, E, stream);
}
int ccd_patch_embed_bwd_g16(const float* img, const ccd_bf16* g, float* d_w, float* d_bias, float* d_pos, ccd_bf16* ws_patches, int views,
                            int E, void* stream) {
    return ccd_patch_embed_bwd_any(img, g, 1, d_w, d_bias, d_pos, nullptr, ws_patches, views, E, stream);
}
int ccd_small_matmul_f32(const float* A, const float* B, float* C, int M, int N, int K, int trans_a, int accumulate,
                         void* stream) {
    CCD_CHECK(A && B && C && M > 0 && N > 0 && K > 0, CCD_EINVAL);
    CCD_LAUNCH(ccd::small_matmul_f32_kernel, dim3((N + 127) / 128, M), dim3(128), 0, stream, A, B, C, M, N, K, trans_a,
               accumulate);
    return ccd_rt_last_error();
}
int ccd_colsum_bf16(const ccd_bf16* x, long ld, int rows, int N, const int* d_rows, int rows_mul, float* out,
                    void* stream) {
    CCD_CHECK(x && out, CCD_EINVAL);
    if (rows == 0 || N == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0 && CCD_ALIGNED16(x), CCD_ESHAPE);
    int cgn_log2, col_blocks, rpb, row_blocks;
    ccd_reduce_geometry(rows, N, &cgn_log2, &col_blocks, &rpb, &row_blocks);
    CCD_LAUNCH(ccd::colsum_bf16_kernel, dim3(col_blocks, row_blocks), dim3(ccd::COLSUM_THREADS), 0, stream, x, ld, rows, N, d_rows,
               rows_mul, out, rpb, cgn_log2);
    return ccd_rt_last_error();
}
int ccd_mirror_bf16(const ccd_mirror_desc* d_descs, int ndesc, int total_tiles, void* stream) {
    CCD_CHECK(d_descs && ndesc > 0 && total_tiles > 0, CCD_EINVAL);
    static_assert(sizeof(ccd_mirror_desc) == sizeof(ccd::MirrorDesc) + 4 || sizeof(ccd_mirror_desc) == sizeof(ccd::MirrorDesc),
                  "descriptor layout");
    CCD_LAUNCH(ccd::mirror_bf16_kernel, dim3(total_tiles), dim3(256), 0, stream,
               reinterpret_cast<const ccd::MirrorDesc*>(d_descs), ndesc);
    return ccd_rt_last_error();
}
int ccd_scale_cast_rows(const float* src, ccd_bf16* dst, const float* rowscale, int rows_per_sample, long rows, int E,
                        void* stream) {
    CCD_CHECK(src && dst && rows >= 0 && E > 0 && E % 4 == 0 && rows_per_sample > 0, CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    const long n = rows * (E / 4);
    CCD_LAUNCH(ccd::scale_cast_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, rowscale,
               rows_per_sample, rows, E);
    return ccd_rt_last_error();
}
int ccd_cast_bf16(const float* src, ccd_bf16* dst, long n, void* stream) {
    CCD_CHECK(src && dst && n >= 0, CCD_EINVAL);
    if (n == 0) return CCD_OK;
    CCD_LAUNCH(ccd::cast_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, src, dst, n);
    return ccd_rt_last_error();
}

// ------------------------------------------------------------------------------- character-region path
int ccd_ccl_label(const float* mask, uint8_t* idmap, int images, void* stream) {
    CCD_CHECK(mask && idmap && images >= 0, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    CCD_LAUNCH(ccd::ccl_label_kernel, dim3(images), dim3(256), 0, stream, mask, idmap, images);
    return ccd_rt_last_error();
}
int ccd_mask_to_idmap(const float* mask, uint8_t* idmap, int images, void* stream) {
    CCD_CHECK(mask && idmap && images >= 0, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    const long n = (long)images * ccd::CM_PIX;
    CCD_LAUNCH(ccd::mask_to_idmap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mask, idmap, n);
    return ccd_rt_last_error();
}
int ccd_seg_to_mask(const float* seg_logits, float* mask, int images, void* stream) {
    CCD_CHECK(seg_logits && mask && images >= 0, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    const long n = (long)images * ccd::CM_PIX;
    CCD_LAUNCH(ccd::seg_to_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, seg_logits, mask, images);
    return ccd_rt_last_error();
}
int ccd_kmeans2_mask(const uint8_t* gray, const long* offsets, const int* hw, uint8_t* mask, int images, void* stream) {
    CCD_CHECK(gray && offsets && hw && mask && images >= 0, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    CCD_LAUNCH(ccd::kmeans2_mask_kernel, dim3(images), dim3(ccd::KM_THREADS), 0, stream, gray, offsets, hw, mask);
    return ccd_rt_last_error();
}
int ccd_augment_views(const uint8_t* img, const float* params, const float* theta, float* out, uint8_t* staged_ws, int batch,
                      int height, int width, const float* mean3, const float* std3, const uint16_t* overlay, int overlay_layers,
                      const float* warp_maps, int warp_count, void* stream) {
    CCD_CHECK(img && params && theta && out && staged_ws && mean3 && std3 && batch >= 0, CCD_EINVAL);
    CCD_CHECK(overlay_layers >= 0 && (overlay != nullptr) == (overlay_layers > 0), CCD_EINVAL);
    CCD_CHECK(warp_count >= 0 && (warp_maps != nullptr) == (warp_count > 0), CCD_EINVAL);
    CCD_CHECK(height >= 2 && width >= 2 && (long)height * width < (1L << 24), CCD_ESHAPE);
    CCD_CHECK(ccd::aug_spatial_smem(height, width) <= 160 * 1024, CCD_ESHAPE);      // the pre-pass keeps a (sample, view) image in LDS
    CCD_CHECK(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, CCD_EINVAL);
    if (batch == 0) return CCD_OK;
    CCD_LAUNCH(ccd::augment_spatial_kernel, dim3(2 * batch), dim3(256), (int)ccd::aug_spatial_smem(height, width), stream, img, params,
               staged_ws, height, width, overlay, overlay_layers);
    CCD_LAUNCH(ccd::augment_views_kernel, dim3((height * width + 255) / 256, batch), dim3(256), 0, stream, img, staged_ws, theta,
               out, batch, height, width, mean3[0], mean3[1], mean3[2], 1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2], params, warp_maps,
               warp_count);
    return ccd_rt_last_error();
}
int ccd_warp_idmap(const uint8_t* src, const float* theta, int theta_stride, uint8_t* dst, int images, void* stream) {
    CCD_CHECK(src && theta && dst && images >= 0 && theta_stride >= 6, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    CCD_LAUNCH(ccd::warp_idmap_kernel, dim3(ccd::CM_PIX / 256, images), dim3(256), 0, stream, src, theta, theta_stride,
               dst, images);
    return ccd_rt_last_error();
}
int ccd_region_stats(const uint8_t* idmap, uint8_t* tok_plane, float* tok_coef, uint8_t* present, int views,
                     void* stream) {
    CCD_CHECK(idmap && tok_plane && tok_coef && present && views >= 0, CCD_EINVAL);
    if (views == 0) return CCD_OK;
    CCD_LAUNCH(ccd::region_stats_kernel, dim3(views), dim3(256), 0, stream, idmap, tok_plane, tok_coef, present, views);
    return ccd_rt_last_error();
}
int ccd_select_scan(const uint8_t* present, int batch, int* nsel, int* offset, int* total, uint8_t* new_index,
                    void* stream) {
    CCD_CHECK(present && nsel && offset && total && new_index && batch > 0, CCD_EINVAL);
    CCD_LAUNCH(ccd::select_scan_kernel, dim3(1), dim3(256), 0, stream, present, batch, nsel, offset, total, new_index);
    return ccd_rt_last_error();
}
int ccd_region_pool_fwd(const ccd_bf16* feat, const uint8_t* tok_plane, const float* tok_coef, const int* nsel,
                        const int* offset, const int* total, ccd_bf16* rows, int batch, int E, void* stream) {
    CCD_CHECK(feat && tok_plane && tok_coef && nsel && offset && total && rows && batch > 0 && E > 0, CCD_EINVAL);
    const size_t smem = (size_t)ccd::CM_PLANES * E * 4;
    CCD_CHECK(smem <= 120 * 1024 && E % 2 == 0, CCD_ESHAPE);
    CCD_LAUNCH(ccd::region_pool_fwd_kernel, dim3(2 * batch), dim3(256), smem, stream, feat, tok_plane, tok_coef, nsel,
               offset, total, rows, batch, E);
    return ccd_rt_last_error();
}
int ccd_region_pool_bwd(const ccd_bf16* d_rows, const uint8_t* tok_plane, const float* tok_coef, const int* nsel,
                        const int* offset, const int* total, ccd_bf16* d_feat, int batch, int E, void* stream) {
    CCD_CHECK(d_rows && tok_plane && tok_coef && nsel && offset && total && d_feat && batch > 0 && E > 0, CCD_EINVAL);
    CCD_CHECK(E % 8 == 0, CCD_ESHAPE);
    CCD_LAUNCH(ccd::region_pool_bwd_kernel, dim3(2 * batch), dim3(256), 0, stream, d_rows, tok_plane, tok_coef, nsel,
               offset, total, d_feat, batch, E);
    return ccd_rt_last_error();
}
int ccd_idmap_to_planes(const uint8_t* idmap, float* planes, int images, void* stream) {
    CCD_CHECK(idmap && planes && images >= 0, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    const long n = (long)images * ccd::CM_PLANES * ccd::CM_PIX;
    CCD_LAUNCH(ccd::idmap_to_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, idmap, planes,
               (long)images);
    return ccd_rt_last_error();
}
int ccd_planes_to_idmap(const float* planes, uint8_t* idmap, int images, void* stream) {
    CCD_CHECK(idmap && planes && images >= 0, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    const long n = (long)images * ccd::CM_PIX;
    CCD_LAUNCH(ccd::planes_to_idmap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, planes, idmap,
               (long)images);
    return ccd_rt_last_error();
}

// ------------------------------------------------------------------------------- DINO head pieces
int ccd_l2norm_fwd(const ccd_bf16* x, ccd_bf16* y, float* inv, int max_rows, const int* d_rows, int rows_mul, int D,
                   void* stream) {
    CCD_CHECK(x && y && inv && max_rows >= 0, CCD_EINVAL);
    if (max_rows == 0) return CCD_OK;
    CCD_CHECK(D > 0 && D <= 64 * ccd::HD_MAX_PER_LANE, CCD_ESHAPE);
    CCD_LAUNCH(ccd::l2norm_fwd_kernel, dim3((max_rows + 3) / 4), dim3(256), 0, stream, x, y, inv, max_rows, d_rows,
               rows_mul, D);
    return ccd_rt_last_error();
}
int ccd_l2norm_bwd(const ccd_bf16* x, const float* inv, const ccd_bf16* dy, ccd_bf16* dx, int max_rows,
                   const int* d_rows, int rows_mul, int D, void* stream) {
    CCD_CHECK(x && inv && dy && dx && max_rows >= 0, CCD_EINVAL);
    if (max_rows == 0) return CCD_OK;
    CCD_CHECK(D > 0 && D <= 64 * ccd::HD_MAX_PER_LANE, CCD_ESHAPE);
    CCD_LAUNCH(ccd::l2norm_bwd_kernel, dim3((max_rows + 3) / 4), dim3(256), 0, stream, x, inv, dy, dx, max_rows, d_rows,
               rows_mul, D);
    return ccd_rt_last_error();
}
int ccd_weightnorm_fwd(const float* v, const float* g, ccd_bf16* w, ccd_bf16* w_t, float* inv, int K, int D,
                       void* stream) {
    CCD_CHECK(v && g && w && inv && K > 0 && D > 0, CCD_EINVAL);
    const size_t smem = (size_t)ccd::WN_ROWS * (D + 1) * 4;
    CCD_CHECK(smem <= 128 * 1024 && D % 4 == 0 && D <= 256 * ccd::WN_MAX_V4 && K % 2 == 0, CCD_ESHAPE);
    CCD_LAUNCH(ccd::weightnorm_fwd_kernel, dim3((K + ccd::WN_ROWS - 1) / ccd::WN_ROWS), dim3(256), smem, stream, v, g, w, w_t, inv, K, D);
    return ccd_rt_last_error();
}
int ccd_weightnorm_bwd(const float* v, const float* g, const float* inv, const float* dw, float* dv, float* dg, int K,
                       int D, void* stream) {
    CCD_CHECK(v && g && inv && dw && dv && K > 0 && D > 0, CCD_EINVAL);
    CCD_LAUNCH(ccd::weightnorm_bwd_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, v, g, inv, dw, dv, dg, K, D);
    return ccd_rt_last_error();
}

// ------------------------------------------------------------------------------- losses
int ccd_dino_loss_fwd(const float* s_logits, const float* t_logits, const float* center, int K, const int* d_m,
                      int max_rows, float student_temp, float teacher_temp, float* stats, float* loss_out,
                      void* stream) {
    CCD_CHECK(s_logits && t_logits && center && d_m && stats && loss_out, CCD_EINVAL);
    CCD_CHECK(K > 0 && K % 4 == 0 && max_rows > 0 && student_temp > 0 && teacher_temp > 0, CCD_ESHAPE);
    CCD_LAUNCH(ccd::dino_loss_fwd_kernel, dim3(max_rows), dim3(256), 0, stream, s_logits, t_logits, center, K, d_m,
               max_rows, 1.0f / student_temp, 1.0f / teacher_temp, stats, loss_out);
    return ccd_rt_last_error();
}
int ccd_dino_loss_bwd(const float* s_logits, const float* t_logits, const float* center, int K, const int* d_m,
                      int max_rows, float student_temp, float teacher_temp, const float* stats, float grad_scale,
                      const float* d_grad_scale, ccd_bf16* d_logits, void* stream) {
    CCD_CHECK(s_logits && t_logits && center && d_m && stats && d_logits, CCD_EINVAL);
    CCD_CHECK(K > 0 && K % 4 == 0 && max_rows > 0 && student_temp > 0 && teacher_temp > 0, CCD_ESHAPE);
    CCD_LAUNCH(ccd::dino_loss_bwd_kernel, dim3(max_rows), dim3(256), 0, stream, s_logits, t_logits, center, K, d_m,
               max_rows, 1.0f / student_temp, 1.0f / teacher_temp, stats, grad_scale, d_grad_scale, d_logits);
    return ccd_rt_last_error();
}
// head's last layer + distillation loss, logits never written (headloss.h).  Column splits: a multiple of 8 (one residue per XCD),
// 1 024 columns per split where K allows it
static int ccd_head_loss_splits(int K) {
    if (K <= 0 || K % 512 != 0) return 0;
    const int chunks_total = K / 64;
    for (int cs = 64; cs >= 8; cs -= 8)
        if (chunks_total % cs == 0) return cs;
    return 0;
}
long ccd_head_loss_ws_floats(int max_rows, int K) {
    const int cs = ccd_head_loss_splits(K);
    return cs > 0 && max_rows > 0 ? (long)cs * max_rows * 8 : 0;
}
static int ccd_head_loss_launch(ccd::HeadLossParams& q, bool bwd, void* stream) {
    q.CS = ccd_head_loss_splits(q.K);
    CCD_CHECK(q.CS > 0, CCD_ESHAPE);
    q.chunks = q.K / 64 / q.CS;
    const int smem = ccd::hl_smem_bytes(q.chunks);
    CCD_CHECK(smem <= 160 * 1024, CCD_ESHAPE);
    int grid = (2 * ccd_grid_cus()) & ~7;          // two workgroups per CU (<= 256 registers, 72 KiB of LDS): one's softmax update runs under the other's products
    if (grid < 8) grid = 8;
    if (bwd) CCD_LAUNCH((ccd::head_loss_kernel<true>), dim3(grid), dim3(ccd::HL_THREADS), smem, stream, q);
    else CCD_LAUNCH((ccd::head_loss_kernel<false>), dim3(grid), dim3(ccd::HL_THREADS), smem, stream, q);
    return ccd_rt_last_error();
}
static int ccd_head_loss_fill(ccd::HeadLossParams& q, const ccd_bf16* zs, long ld_zs, const ccd_bf16* zt, long ld_zt, const ccd_bf16* ws,
                              long ld_ws, const ccd_bf16* wt, long ld_wt, const float* center, int K, int D, const int* d_m, int max_rows,
                              float student_temp, float teacher_temp) {
    CCD_CHECK(zs && zt && ws && wt && center && d_m, CCD_EINVAL);
    CCD_CHECK(D == ccd::HL_D && K > 0 && max_rows > 0 && student_temp > 0 && teacher_temp > 0, CCD_ESHAPE);
    CCD_CHECK(CCD_ALIGNED16(zs) && CCD_ALIGNED16(zt) && CCD_ALIGNED16(ws) && CCD_ALIGNED16(wt) && CCD_ALIGNED16(center), CCD_EINVAL);
    CCD_CHECK(ld_zs % 8 == 0 && ld_zt % 8 == 0 && ld_ws % 8 == 0 && ld_wt % 8 == 0 && ld_zs >= D && ld_zt >= D && ld_ws >= D && ld_wt >= D, CCD_EINVAL);
    CCD_CHECK(((long)max_rows + 1) * ld_zs * 2 < CCD_MAX_OPERAND_BYTES && ((long)max_rows + 1) * ld_zt * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    // (the weight matrices are walked by 64-bit pointer + a per-lane 32-bit row offset of at most 64 rows)
    q.zs = reinterpret_cast<const ccd::bf16_t*>(zs); q.ld_zs = ld_zs; q.zt = reinterpret_cast<const ccd::bf16_t*>(zt); q.ld_zt = ld_zt;
    q.ws = reinterpret_cast<const ccd::bf16_t*>(ws); q.ld_ws = ld_ws; q.wt = reinterpret_cast<const ccd::bf16_t*>(wt); q.ld_wt = ld_wt;
    q.center = center; q.d_m = d_m; q.max_rows = max_rows; q.K = K;
    q.ks = 1.4426950408889634f / student_temp; q.kt = 1.4426950408889634f / teacher_temp;
    q.part = nullptr; q.stats = nullptr; q.grad_scale = 0.f; q.d_grad_scale = nullptr; q.d_logits = nullptr; q.ld_d = 0;
    return CCD_OK;
}
int ccd_head_loss_fwd(const ccd_bf16* zs, long ld_zs, const ccd_bf16* zt, long ld_zt, const ccd_bf16* ws, long ld_ws,
                      const ccd_bf16* wt, long ld_wt, const float* center, int K, int D, const int* d_m, int max_rows,
                      float student_temp, float teacher_temp, float* ws_part, float* stats, float* loss_out, void* stream) {
    ccd::HeadLossParams q;
    const int rc = ccd_head_loss_fill(q, zs, ld_zs, zt, ld_zt, ws, ld_ws, wt, ld_wt, center, K, D, d_m, max_rows, student_temp, teacher_temp);
    if (rc != CCD_OK) return rc;
    CCD_CHECK(ws_part && stats && loss_out && CCD_ALIGNED16(ws_part) && CCD_ALIGNED16(stats), CCD_EINVAL);
    q.part = ws_part;
    const int rl = ccd_head_loss_launch(q, false, stream);
    if (rl != CCD_OK) return rl;
    CCD_LAUNCH(ccd::head_loss_finish_kernel, dim3((max_rows + 255) / 256), dim3(256), 0, stream, (const float*)ws_part, d_m, max_rows, q.CS, stats, loss_out);
    return ccd_rt_last_error();
}
int ccd_head_loss_bwd(const ccd_bf16* zs, long ld_zs, const ccd_bf16* zt, long ld_zt, const ccd_bf16* ws, long ld_ws,
                      const ccd_bf16* wt, long ld_wt, const float* center, int K, int D, const int* d_m, int max_rows,
                      float student_temp, float teacher_temp, const float* stats, float grad_scale, const float* d_grad_scale,
                      ccd_bf16* d_logits, long ld_d, void* stream) {
    ccd::HeadLossParams q;
    const int rc = ccd_head_loss_fill(q, zs, ld_zs, zt, ld_zt, ws, ld_ws, wt, ld_wt, center, K, D, d_m, max_rows, student_temp, teacher_temp);
    if (rc != CCD_OK) return rc;
    CCD_CHECK(stats && d_logits && CCD_ALIGNED16(stats) && CCD_ALIGNED16(d_logits) && ld_d % 8 == 0 && ld_d >= K, CCD_EINVAL);
    CCD_CHECK(33L * ld_d * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    q.stats = stats; q.grad_scale = grad_scale; q.d_grad_scale = d_grad_scale;
    q.d_logits = reinterpret_cast<ccd::bf16_t*>(d_logits); q.ld_d = ld_d;
    return ccd_head_loss_launch(q, true, stream);
}
int ccd_colsum_f32(const float* x, int K, const int* d_rows, int rows_mul, int max_rows, float* out, void* stream) {
    CCD_CHECK(x && out && K > 0 && K % 4 == 0 && max_rows > 0, CCD_EINVAL);
    const int col_blocks = (K / 4 + 255) / 256;
    int row_blocks = (8 * ccd_rt_num_cus() + col_blocks - 1) / col_blocks;
    if (row_blocks > max_rows) row_blocks = max_rows;
    const int rpb = (max_rows + row_blocks - 1) / row_blocks;
    row_blocks = (max_rows + rpb - 1) / rpb;
    CCD_LAUNCH(ccd::colsum_f32_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, stream, x, K, d_rows, rows_mul,
               max_rows, rpb, out);
    return ccd_rt_last_error();
}
int ccd_matvec_bf16(const ccd_bf16* w, long ldw, const float* v, int K, int D, float* out, void* stream) {
    CCD_CHECK(w && v && out && K > 0 && D > 0 && D % 256 == 0 && ldw % 8 == 0 && CCD_ALIGNED16(w) && CCD_ALIGNED16(v), CCD_EINVAL);
    int blocks = (K + 7) / 8;                       // a block = 4 waves = 8 rows per trip
    const int cap = 4 * ccd_rt_num_cus();
    if (blocks > cap) blocks = cap;
    CCD_LAUNCH(ccd::matvec_bf16_kernel, dim3(blocks), dim3(256), 0, stream, w, ldw, v, K, D, out);
    return ccd_rt_last_error();
}
int ccd_center_ema(float* center, const float* batch_sum, int K, const int* d_m, int world, float momentum,
                   void* stream) {
    CCD_CHECK(center && batch_sum && d_m && K > 0 && world > 0, CCD_EINVAL);
    CCD_LAUNCH(ccd::center_ema_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, center, batch_sum, K, d_m, world,
               momentum);
    return ccd_rt_last_error();
}
int ccd_seg_loss(const float* logits, const float* mask_a, const uint8_t* idmap_b, int half, float grad_scale,
                 float* loss_out, float* d_logits, void* stream) {
    CCD_CHECK(logits && mask_a && idmap_b && loss_out && half > 0, CCD_EINVAL);
    const long npix = 2L * half * ccd::CM_PIX;
    const long per_block = 256L * ccd::SEG_LOSS_PER_THREAD;
    CCD_LAUNCH(ccd::seg_loss_kernel, dim3((unsigned)((npix + per_block - 1) / per_block)), dim3(256), 0, stream, logits, mask_a, idmap_b,
               half, grad_scale, loss_out, d_logits);
    return ccd_rt_last_error();
}

// ------------------------------------------------------------------------------- optimiser
int ccd_seg_sumsq(const float* grad, const int* chunk_seg, const long* chunk_begin, const int* chunk_len, int nchunks,
                  float* norm2, void* stream) {
    CCD_CHECK(grad && chunk_seg && chunk_begin && chunk_len && norm2 && nchunks > 0, CCD_EINVAL);
    const int cpb = 32;                                     // 128 KiB of gradients per workgroup
    CCD_LAUNCH(ccd::seg_sumsq_kernel, dim3((nchunks + cpb - 1) / cpb), dim3(256), 0, stream, grad, chunk_seg, chunk_begin,
               chunk_len, norm2, nchunks, cpb);
    return ccd_rt_last_error();
}
int ccd_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, ccd_bf16* mirror,
              const int* chunk_seg, const long* chunk_begin, const int* chunk_len, int nchunks,
              const ccd_seg_hyper* hyper, const float* norm2, float clip, float beta1, float beta2, float eps,
              void* stream) {
    CCD_CHECK(param && grad && exp_avg && exp_avg_sq && chunk_seg && chunk_begin && chunk_len && hyper && norm2, CCD_EINVAL);
    CCD_CHECK(nchunks > 0, CCD_EINVAL);
    static_assert(sizeof(ccd_seg_hyper) == sizeof(ccd::SegHyper), "hyper layout");
    CCD_LAUNCH(ccd::adamw_kernel, dim3(nchunks), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, mirror, chunk_seg,
               chunk_begin, chunk_len, reinterpret_cast<const ccd::SegHyper*>(hyper), norm2, clip, beta1, beta2, eps);
    return ccd_rt_last_error();
}
int ccd_clip_scale(float* grad, const int* chunk_seg, const long* chunk_begin, const int* chunk_len, int nchunks,
                   const float* norm2, float clip, void* stream) {
    CCD_CHECK(grad && chunk_seg && chunk_begin && chunk_len && norm2 && nchunks > 0 && clip > 0, CCD_EINVAL);
    CCD_LAUNCH(ccd::clip_scale_kernel, dim3(nchunks), dim3(256), 0, stream, grad, chunk_seg, chunk_begin, chunk_len, norm2,
               clip);
    return ccd_rt_last_error();
}
int ccd_ema(float* teacher, const float* student, ccd_bf16* mirror, long n, float m, float one_minus_m, const float* d_m,
            void* stream) {
    CCD_CHECK(teacher && student && n >= 0, CCD_EINVAL);
    if (n == 0) return CCD_OK;
    CCD_LAUNCH(ccd::ema_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, teacher, student, mirror, n, m,
               one_minus_m, d_m);
    return ccd_rt_last_error();
}

// --------------------------------------------------------------------------------- segmentation head
static int ccd_check_conv_desc(const ccd_conv_desc* d) {
    CCD_CHECK(d, CCD_EINVAL);
    CCD_CHECK(d->ntaps >= 1 && d->ntaps <= 16 && d->cin > 0 && d->cin % 64 == 0, CCD_ESHAPE);
    CCD_CHECK(d->g_h_log2 >= 0 && d->g_w_log2 >= 0 && d->g_h_log2 + d->g_w_log2 <= 24, CCD_ESHAPE);
    CCD_CHECK(d->s_h > 0 && d->s_w > 0 && d->s_mul >= 1, CCD_ESHAPE);
    for (int i = 0; i < d->ntaps; ++i) CCD_CHECK(d->dy[i] >= -8 && d->dy[i] <= 7 && d->dx[i] >= -8 && d->dx[i] <= 7, CCD_ESHAPE);
    return CCD_OK;
}

static void ccd_fill_gather(ccd::GemmParams& p, const ccd_conv_desc* desc) {
    p.g_h_log2 = desc->g_h_log2; p.g_w_log2 = desc->g_w_log2; p.s_h = desc->s_h; p.s_w = desc->s_w;
    p.s_mul = desc->s_mul; p.cin = desc->cin;
    p.dy_pack = 0; p.dx_pack = 0;
    for (int i = 0; i < 16; ++i) {
        p.dy_pack |= (unsigned long long)((desc->dy[i] + 8) & 15) << (4 * i);
        p.dx_pack |= (unsigned long long)((desc->dx[i] + 8) & 15) << (4 * i);
    }
}

int ccd_conv_gemm(const ccd_bf16* src, long src_ld, const ccd_conv_desc* desc, const ccd_bf16* W, long ldw, int M, int N,
                  ccd_bf16* C, long ldc, const float* bias, float* colsum, float* colsumsq, void* stream) {
    CCD_CHECK(src && W && C, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(src) && CCD_ALIGNED16(W) && CCD_ALIGNED16(C), CCD_EINVAL);
    const int rc = ccd_check_conv_desc(desc);
    if (rc != CCD_OK) return rc;
    if (M == 0 || N == 0) return CCD_OK;
    CCD_CHECK(M > 0 && N > 0, CCD_EINVAL);
    CCD_CHECK(N % 8 == 0 && src_ld % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && src_ld >= desc->cin, CCD_ESHAPE);
    CCD_CHECK(M % (1 << (desc->g_h_log2 + desc->g_w_log2)) == 0, CCD_ESHAPE);
    {
        const long images = M >> (desc->g_h_log2 + desc->g_w_log2);
        CCD_CHECK(images * desc->s_h * desc->s_w * src_ld * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
        CCD_CHECK(((long)N * ldw + (long)desc->ntaps * desc->cin) * 2 < CCD_MAX_OPERAND_BYTES, CCD_ESHAPE);
    }
    ccd::GemmParams p = ccd::GemmParams();
    p.A = src; p.lda = src_ld; p.B = W; p.ldb = ldw; p.M = M; p.N = N; p.K = desc->ntaps * desc->cin;
    p.C = C; p.ldc = ldc; p.bias = bias; p.rows_per_sample = 1; p.k_per_split = p.K; p.alpha = 1.0f; p.rows_mul = 1;
    p.colsum = colsum; p.colsumsq = colsumsq;
    ccd_fill_gather(p, desc);
    p.c_map = desc->c_map; p.c_py = desc->c_py; p.c_px = desc->c_px;
    const int tiles = ((M + ccd::GEMM_BM - 1) / ccd::GEMM_BM) * ((N + ccd::GEMM_BN - 1) / ccd::GEMM_BN);
    CCD_LAUNCH((ccd::gemm_bf16_kernel<false, ccd::EPI_BF16, true>), dim3(ccd_gemm_grid(p, tiles, 1)), dim3(256),
               (size_t)ccd::GEMM_SMEM_BYTES, stream, p);
    return ccd_rt_last_error();
}

int ccd_conv_wgrad(const ccd_bf16* A, long lda, int P, const ccd_bf16* src, long src_ld, const ccd_conv_desc* desc,
                   long rows, float* out, long ldo, void* stream) {
    CCD_CHECK(A && src && out, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(A) && CCD_ALIGNED16(src) && CCD_ALIGNED16(out), CCD_EINVAL);
    const int rc = ccd_check_conv_desc(desc);
    if (rc != CCD_OK) return rc;
    if (rows == 0 || P == 0) return CCD_OK;
    const int Q = desc->ntaps * desc->cin;
    CCD_CHECK(rows > 0 && rows < (1L << 31) && P > 0, CCD_EINVAL);
    CCD_CHECK(P % 8 == 0 && lda % 8 == 0 && src_ld % 8 == 0 && ldo % 4 == 0 && src_ld >= desc->cin, CCD_ESHAPE);
    CCD_CHECK(desc->g_w_log2 >= 2 && rows % (1 << (desc->g_h_log2 + desc->g_w_log2)) == 0, CCD_ESHAPE);
    CCD_CHECK((rows >> (desc->g_h_log2 + desc->g_w_log2)) * desc->s_h * desc->s_w * src_ld * 2 < CCD_MAX_OPERAND_BYTES,
              CCD_ESHAPE);
    const int tiles = ((P + 127) / 128) * ((Q + 127) / 128);
    int splits = (2 * ccd_rt_num_cus()) / tiles;
    if (splits < 1) splits = 1;
    int per = (int)((rows + splits - 1) / splits);
    per = ((per + 63) / 64) * 64;
    splits = (int)((rows + per - 1) / per);
    ccd::GemmParams p = ccd::GemmParams();
    p.A = A; p.lda = lda; p.B = src; p.ldb = src_ld; p.M = P; p.N = Q; p.K = (int)rows;
    p.C = out; p.ldc = ldo; p.rows_per_sample = 1; p.k_per_split = per; p.alpha = 1.0f; p.rows_mul = 1;
    ccd_fill_gather(p, desc);
    CCD_LAUNCH((ccd::gemm_bf16_kernel<true, ccd::EPI_ATOMIC, true>), dim3(ccd_gemm_grid(p, tiles, splits)), dim3(256),
               (size_t)ccd::GEMM_SMEM_BYTES, stream, p);
    return ccd_rt_last_error();
}

int ccd_im2col(const ccd_bf16* src, long src_ld, const ccd_conv_desc* desc, long rows, ccd_bf16* cols, void* stream) {
    CCD_CHECK(src && cols && CCD_ALIGNED16(src) && CCD_ALIGNED16(cols), CCD_EINVAL);
    const int rc = ccd_check_conv_desc(desc);
    if (rc != CCD_OK) return rc;
    if (rows == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && src_ld % 8 == 0 && src_ld >= desc->cin, CCD_ESHAPE);
    const long total = rows * desc->ntaps * (desc->cin / 8);
    CCD_LAUNCH(ccd::im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, src_ld, *desc, rows,
               cols);
    return ccd_rt_last_error();
}

int ccd_bn_finalize(const float* stats, float count, float eps, float momentum, float* mean_rstd, float* running_mean,
                    float* running_var, int C, void* stream) {
    CCD_CHECK(stats && mean_rstd && running_mean && running_var && C > 0 && count > 1.0f, CCD_EINVAL);
    CCD_LAUNCH(ccd::bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, stats, count, eps, momentum, mean_rstd,
               running_mean, running_var, C);
    return ccd_rt_last_error();
}

int ccd_bn_relu_fwd(const ccd_bf16* x, long ldx, const float* mean_rstd, const float* gamma, const float* beta,
                    ccd_bf16* y, long ldy, long rows, int C, void* stream) {
    CCD_CHECK(x && mean_rstd && gamma && beta && y && CCD_ALIGNED16(x) && CCD_ALIGNED16(y), CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, CCD_ESHAPE);
    CCD_CHECK(C <= 256, CCD_ESHAPE);                          // (the grid-stride keeps a thread on its channels: C / 8 divides 256)
    const long total = rows * (C / 8);
    CCD_LAUNCH(ccd::bn_relu_fwd_kernel, dim3(ccd_stream_blocks(total, 4)), dim3(256), 0, stream, x, ldx, mean_rstd,
               gamma, beta, y, ldy, rows, C);
    return ccd_rt_last_error();
}

int ccd_bn_relu_bwd_reduce(const ccd_bf16* dy, long lddy, const ccd_bf16* x, long ldx, const float* mean_rstd,
                           const float* gamma, const float* beta, float* red, long rows, int C, void* stream) {
    CCD_CHECK(dy && x && mean_rstd && gamma && beta && red && CCD_ALIGNED16(dy) && CCD_ALIGNED16(x), CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0, CCD_ESHAPE);
    int cgn_log2, col_blocks, rpb, row_blocks;
    ccd_reduce_geometry(rows, C, &cgn_log2, &col_blocks, &rpb, &row_blocks);
    CCD_LAUNCH(ccd::bn_relu_bwd_reduce_kernel, dim3(col_blocks, row_blocks), dim3(ccd::COLSUM_THREADS), 0, stream, dy, lddy, x, ldx,
               mean_rstd, gamma, beta, red, rows, C, rpb, cgn_log2);
    return ccd_rt_last_error();
}

int ccd_bn_relu_bwd_apply(const ccd_bf16* dy, long lddy, const ccd_bf16* x, long ldx, const float* mean_rstd,
                          const float* gamma, const float* beta, const float* red, float count, const float* red_local,
                          float* dgamma, float* dbeta, ccd_bf16* dx, long lddx, long rows, int C, void* stream) {
    CCD_CHECK(dy && x && mean_rstd && gamma && beta && red && red_local && dgamma && dbeta && dx, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(dy) && CCD_ALIGNED16(x) && CCD_ALIGNED16(dx), CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && C > 0 && C % 8 == 0 && C <= 256 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, CCD_ESHAPE);
    const long total = rows * (C / 8);
    CCD_LAUNCH(ccd::bn_relu_bwd_apply_kernel, dim3(ccd_stream_blocks(total, 2)), dim3(256), 0, stream, dy, lddy, x,
               ldx, mean_rstd, gamma, beta, red, count, red_local, dgamma, dbeta, dx, lddx, rows, C);
    return ccd_rt_last_error();
}

int ccd_cls_gather_fwd(const float* zT, long ldz, const float* bias, float* logits, int images, int H, int W,
                       void* stream) {
    CCD_CHECK(zT && bias && logits, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    CCD_CHECK(images > 0 && H > 0 && W > 0 && ldz >= (long)images * H * W, CCD_ESHAPE);
    const long total = (long)images * H * W;
    CCD_LAUNCH(ccd::cls_gather_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, zT, ldz, bias,
               logits, images, H, W);
    return ccd_rt_last_error();
}

int ccd_cls_grad_cols(const float* dlogits, ccd_bf16* g, int images, int H, int W, void* stream) {
    CCD_CHECK(dlogits && g && CCD_ALIGNED16(g), CCD_EINVAL);
    if (images == 0) return CCD_OK;
    CCD_CHECK(images > 0 && H > 0 && W > 0, CCD_ESHAPE);
    const long total = (long)images * H * W * 8;
    CCD_LAUNCH(ccd::cls_grad_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dlogits, g, images,
               H, W);
    return ccd_rt_last_error();
}

static bool ccd_cls_tail_shape(int images, int H, int W, int C, long ld) {
    return images > 0 && H == ccd::CT_H && W == ccd::CT_W && C == ccd::CT_C && ld >= C && ld % 8 == 0;
}

int ccd_cls_tail_fwd(const ccd_bf16* y, long ldy, const float* mean_rstd, const float* gamma, const float* beta,
                     const float* w, const float* bias, float* logits, int images, int H, int W, int C, void* stream) {
    CCD_CHECK(y && mean_rstd && gamma && beta && w && bias && logits && CCD_ALIGNED16(y), CCD_EINVAL);
    if (images == 0) return CCD_OK;
    CCD_CHECK(ccd_cls_tail_shape(images, H, W, C, ldy), CCD_ESHAPE);
    // a band re-derives the z rows above and below it (8 rows: + 25 % of y read twice, 16: + 12 %): whole images once they fill the chip
    // twice over
    const long cus = ccd_grid_cus();
    const int band_rows = images >= 2 * cus ? 32 : images * 2 >= 2 * cus ? 16 : 8;
    CCD_LAUNCH(ccd::cls_tail_fwd_kernel, dim3((unsigned)(images * (ccd::CT_H / band_rows))), dim3(256), 0, stream, y, ldy,
               mean_rstd, gamma, beta, w, bias, logits, band_rows);
    return ccd_rt_last_error();
}

static unsigned ccd_cls_tail_grid(int nbands, int per_cu) {
    const long cap = (long)per_cu * ccd_grid_cus();          // (persistent grids leave the reserved CUs to an attached gradient reducer)
    return (unsigned)(nbands < cap ? nbands : cap);
}

int ccd_cls_tail_bwd_reduce(const float* dlogits, const ccd_bf16* y, long ldy, const float* mean_rstd, const float* gamma,
                            const float* beta, const float* w, float* red, float* db_cls, int images, int H, int W, int C,
                            void* stream) {
    CCD_CHECK(dlogits && y && mean_rstd && gamma && beta && w && red && db_cls && CCD_ALIGNED16(y), CCD_EINVAL);
    if (images == 0) return CCD_OK;
    CCD_CHECK(ccd_cls_tail_shape(images, H, W, C, ldy), CCD_ESHAPE);
    // (two workgroups per CU is what the kernel's registers allow: a third of the grid queued behind them cost 20 %)
    constexpr int RB = 4;
    const int nbands = images * (ccd::CT_H / RB);
    CCD_LAUNCH((ccd::cls_tail_bwd_kernel<false, RB>), dim3(ccd_cls_tail_grid(nbands, 2)), dim3(256), 0, stream, dlogits, y, ldy,
               mean_rstd, gamma, beta, w, red, 1.0f, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
               db_cls, (float*)nullptr, (ccd::bf16_t*)nullptr, 0L, nbands);
    return ccd_rt_last_error();
}

int ccd_cls_tail_bwd_apply(const float* dlogits, const ccd_bf16* y, long ldy, const float* mean_rstd, const float* gamma,
                           const float* beta, const float* w, const float* red, float count, const float* red_local,
                           float* dgamma, float* dbeta, float* dw_cls, float* dbias_t, ccd_bf16* dy, long lddy, int images,
                           int H, int W, int C, void* stream) {
    CCD_CHECK(dlogits && y && mean_rstd && gamma && beta && w && red && red_local && dgamma && dbeta && dw_cls && dbias_t && dy &&
                  CCD_ALIGNED16(y) && CCD_ALIGNED16(dy) && count > 0.f, CCD_EINVAL);
    if (images == 0) return CCD_OK;
    CCD_CHECK(ccd_cls_tail_shape(images, H, W, C, ldy) && lddy >= C && lddy % 8 == 0, CCD_ESHAPE);
    constexpr int RB = 4;
    const int nbands = images * (ccd::CT_H / RB);
    CCD_LAUNCH((ccd::cls_tail_bwd_kernel<true, RB>), dim3(ccd_cls_tail_grid(nbands, 2)), dim3(256), 0, stream, dlogits, y, ldy,
               mean_rstd, gamma, beta, w, const_cast<float*>(red), count, red_local, dgamma, dbeta, dw_cls, (float*)nullptr,
               dbias_t, dy, lddy, nbands);
    return ccd_rt_last_error();
}

int ccd_permute4(const float* src, const long* src_strides, const long* dst_strides, const int* dims, void* dst,
                 int accumulate, void* stream) {
    CCD_CHECK(src && dst && src_strides && dst_strides && dims, CCD_EINVAL);
    ccd::Permute4 q;
    long total = 1;
    for (int i = 0; i < 4; ++i) {
        CCD_CHECK(dims[i] > 0, CCD_EINVAL);
        q.s[i] = src_strides[i]; q.d[i] = dst_strides[i]; q.n[i] = dims[i];
        total *= dims[i];
    }
    const dim3 grid((unsigned)((total + 255) / 256));
    if (accumulate) CCD_LAUNCH((ccd::permute4_kernel<true>), grid, dim3(256), 0, stream, src, q, total, dst);
    else CCD_LAUNCH((ccd::permute4_kernel<false>), grid, dim3(256), 0, stream, src, q, total, dst);
    return ccd_rt_last_error();
}

int ccd_permute4_multi(const ccd_permute4_job* jobs, int n, int accumulate, void* stream) {
    CCD_CHECK(jobs && n >= 0 && n <= ccd::PERMUTE_MULTI_MAX, CCD_EINVAL);
    if (n == 0) return CCD_OK;
    ccd::PermuteJobs t;
    t.n = n;
    unsigned blocks = 0;
    for (int k = 0; k < n; ++k) {
        CCD_CHECK(jobs[k].src && jobs[k].dst, CCD_EINVAL);
        long total = 1;
        for (int i = 0; i < 4; ++i) {
            CCD_CHECK(jobs[k].dims[i] > 0, CCD_EINVAL);
            t.j[k].q.s[i] = jobs[k].src_strides[i]; t.j[k].q.d[i] = jobs[k].dst_strides[i]; t.j[k].q.n[i] = jobs[k].dims[i];
            total *= jobs[k].dims[i];
        }
        t.j[k].src = jobs[k].src; t.j[k].dst = jobs[k].dst; t.j[k].total = total; t.j[k].first_block = blocks;
        blocks += (unsigned)((total + 255) / 256);
    }
    if (accumulate) CCD_LAUNCH((ccd::permute4_multi_kernel<true>), dim3(blocks), dim3(256), 0, stream, t);
    else CCD_LAUNCH((ccd::permute4_multi_kernel<false>), dim3(blocks), dim3(256), 0, stream, t);
    return ccd_rt_last_error();
}

int ccd_bn_finalize_multi(const ccd_bn_finalize_job* jobs, int n, void* stream) {
    CCD_CHECK(jobs && n >= 0 && n <= ccd::BN_MULTI_MAX, CCD_EINVAL);
    if (n == 0) return CCD_OK;
    ccd::BnFinalizeJobs t;
    int cmax = 0;
    for (int k = 0; k < ccd::BN_MULTI_MAX; ++k) {
        const ccd_bn_finalize_job& q = jobs[k < n ? k : 0];
        if (k < n) CCD_CHECK(q.stats && q.mean_rstd && q.running_mean && q.running_var && q.C > 0 && q.count > 1.0f, CCD_EINVAL);
        t.j[k] = ccd::BnFinalizeJob{q.stats, q.mean_rstd, q.running_mean, q.running_var, q.batches, q.count, q.eps, q.momentum, q.C};
        if (k < n && q.C > cmax) cmax = q.C;
    }
    CCD_LAUNCH(ccd::bn_finalize_multi_kernel, dim3((cmax + 63) / 64, n), dim3(64), 0, stream, t);
    return ccd_rt_last_error();
}

// ------------------------------------------------------------------------------------------ finetune path
static void ccd_drop_consts(float p, unsigned* thr, float* scale) {
    if (!(p > 0.f)) { *thr = 0u; *scale = 1.0f; return; }
    const double t = (double)p * 4294967296.0;
    *thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    *scale = 1.0f / (1.0f - p);
}
int ccd_dropout(const void* src, int src_bf16, const float* resid, void* dst, int dst_bf16, long n, uint64_t seed, float p,
                void* stream) {
    CCD_CHECK(n >= 0 && p >= 0.f && p < 1.f, CCD_EINVAL);
    if (n == 0) return CCD_OK;
    CCD_CHECK(src && dst, CCD_EINVAL);
    CCD_CHECK(n % 4 == 0 && CCD_ALIGNED16(resid), CCD_ESHAPE);
    CCD_CHECK((((uintptr_t)src) & (src_bf16 ? 7u : 15u)) == 0 && (((uintptr_t)dst) & (dst_bf16 ? 7u : 15u)) == 0, CCD_EINVAL);
    unsigned thr; float scale;
    ccd_drop_consts(p, &thr, &scale);
    const dim3 grid((unsigned)((n / 4 + 255) / 256)), block(256);
    const unsigned long long sd = seed;
    if (src_bf16 && dst_bf16) CCD_LAUNCH((ccd::dropout_kernel<true, true>), grid, block, 0, stream, src, resid, dst, n, sd, thr, scale);
    else if (src_bf16) CCD_LAUNCH((ccd::dropout_kernel<true, false>), grid, block, 0, stream, src, resid, dst, n, sd, thr, scale);
    else if (dst_bf16) CCD_LAUNCH((ccd::dropout_kernel<false, true>), grid, block, 0, stream, src, resid, dst, n, sd, thr, scale);
    else CCD_LAUNCH((ccd::dropout_kernel<false, false>), grid, block, 0, stream, src, resid, dst, n, sd, thr, scale);
    return ccd_rt_last_error();
}
int ccd_droppath_scales(const float* keep, float* out, int per_block, int nblocks, uint64_t seed, const uint64_t* d_seed,
                        void* stream) {
    CCD_CHECK(keep && out && per_block >= 0 && nblocks >= 0, CCD_EINVAL);
    const long n = (long)per_block * nblocks;
    if (n == 0) return CCD_OK;
    CCD_CHECK(n < (1L << 31), CCD_ESHAPE);
    CCD_LAUNCH(ccd::droppath_scales_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, keep, out, per_block, nblocks,
               (unsigned long long)seed, reinterpret_cast<const unsigned long long*>(d_seed));
    return ccd_rt_last_error();
}
int ccd_dec_embed_fwd(const int64_t* tokens, const float* emb, const float* pos, float* x, int rows, int T, int D,
                      int num_classes, uint64_t seed, float p, void* stream) {
    CCD_CHECK(rows >= 0 && p >= 0.f && p < 1.f, CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(tokens && emb && pos && x, CCD_EINVAL);
    CCD_CHECK(T > 0 && D > 0 && D % 4 == 0 && num_classes > 0, CCD_ESHAPE);
    CCD_CHECK(CCD_ALIGNED16(emb) && CCD_ALIGNED16(pos) && CCD_ALIGNED16(x), CCD_EINVAL);
    unsigned thr; float scale;
    ccd_drop_consts(p, &thr, &scale);
    CCD_LAUNCH(ccd::dec_embed_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const long long*)tokens, emb, pos, x,
               rows, T, D, num_classes, (unsigned long long)seed, thr, scale);
    return ccd_rt_last_error();
}
int ccd_dec_embed_bwd(const int64_t* tokens, const float* dx, float* demb, int rows, int D, int num_classes, int padding_idx,
                      uint64_t seed, float p, void* stream) {
    CCD_CHECK(rows >= 0 && p >= 0.f && p < 1.f, CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(tokens && dx && demb, CCD_EINVAL);
    CCD_CHECK(D > 0 && D <= 1024 && num_classes > 0, CCD_ESHAPE);
    unsigned thr; float scale;
    ccd_drop_consts(p, &thr, &scale);
    const int rows_per_block = 1024;
    CCD_LAUNCH(ccd::dec_embed_bwd_kernel, dim3(num_classes, (rows + rows_per_block - 1) / rows_per_block), dim3(256), 0, stream,
               (const long long*)tokens, dx, demb, rows, D, padding_idx, (unsigned long long)seed, thr, scale, rows_per_block);
    return ccd_rt_last_error();
}
// the MFMA kernels (decoder_xattn.h) cover the unmasked 256-key case; CCD_DEC_ATTN_SIMT=1 forces the general kernels
static bool ccd_dec_attn_mfma(const ccd::DecAttnParams& p) {
    return p.Tk == ccd::XA_TK && !p.tokens && !p.key_len && !p.causal && !ccd_policy().dec_attn_simt;
}
static bool ccd_sattn_mfma(const ccd::DecAttnParams& p) {
    return p.Tk <= ccd::XA_TQ && !ccd_policy().dec_attn_simt;
}
static int ccd_dec_attn_check(const ccd::DecAttnParams& p) {
    CCD_CHECK(p.B >= 0 && p.H > 0, CCD_EINVAL);
    CCD_CHECK(p.Tq > 0 && p.Tq <= ccd::DA_MAXQ && p.Tk > 0 && p.Tk <= ccd::DA_MAXK, CCD_ESHAPE);
    CCD_CHECK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldq >= 64L * p.H && p.ldk >= 64L * p.H &&
              p.ldv >= 64L * p.H && p.ldo >= 64L * p.H, CCD_ESHAPE);
    CCD_CHECK(CCD_ALIGNED16(p.q) && CCD_ALIGNED16(p.k) && CCD_ALIGNED16(p.v), CCD_EINVAL);
    return CCD_OK;
}
int ccd_dec_attn_fwd(const ccd_bf16* q, long ldq, const ccd_bf16* k, long ldk, const ccd_bf16* v, long ldv, ccd_bf16* out,
                     long ldo, float* lse, float* probs, const int64_t* tokens, const int* key_len, int pad_idx, int causal,
                     int B, int H, int Tq, int Tk, float scale, uint64_t seed, float p, void* stream) {
    if (B == 0) return CCD_OK;                               // empty batch: nothing to do (pointers may be null)
    CCD_CHECK(q && k && v && out && lse && p >= 0.f && p < 1.f, CCD_EINVAL);
    ccd::DecAttnParams a = ccd::DecAttnParams();
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.out = out; a.ldo = ldo; a.lse = lse; a.probs = probs;
    a.tokens = (const long long*)tokens; a.key_len = key_len; a.pad_idx = pad_idx; a.causal = causal;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale = scale; a.seed = seed;
    ccd_drop_consts(p, &a.thr, &a.keep_scale);
    const int rc = ccd_dec_attn_check(a);
    if (rc != CCD_OK) return rc;
    if (B == 0) return CCD_OK;
    if (ccd_sattn_mfma(a) && ldo % 8 == 0 && CCD_ALIGNED16(out)) {       // short key sequences: one wave per (sample, head); 16-byte row stores
        CCD_LAUNCH(ccd::sattn_fwd_kernel, dim3((B * H + 3) / 4), dim3(256), (size_t)ccd::SA_FWD_SMEM, stream, a);
        return ccd_rt_last_error();
    }
    if (ccd_dec_attn_mfma(a) && ldo % 8 == 0 && CCD_ALIGNED16(out)) {   // encoder-decoder attention: MFMA kernel
        CCD_LAUNCH(ccd::xattn_fwd_kernel, dim3(B * H), dim3(256), (size_t)ccd::XA_FWD_SMEM, stream, a);
        return ccd_rt_last_error();
    }
    CCD_LAUNCH(ccd::dec_attn_fwd_kernel, dim3(B * H), dim3(256), (size_t)ccd::dec_attn_fwd_smem(Tq, Tk), stream, a);
    return ccd_rt_last_error();
}
int ccd_dec_attn_bwd(const ccd_bf16* q, long ldq, const ccd_bf16* k, long ldk, const ccd_bf16* v, long ldv,
                     const ccd_bf16* out, const ccd_bf16* d_out, long ldo, const float* lse, const int64_t* tokens,
                     const int* key_len, int pad_idx, int causal, int B, int H, int Tq, int Tk, float scale, uint64_t seed,
                     float p, ccd_bf16* dq, long lddq, ccd_bf16* dk, long lddk, ccd_bf16* dv, long lddv, void* stream) {
    if (B == 0) return CCD_OK;
    CCD_CHECK(q && k && v && out && d_out && lse && dq && dk && dv && p >= 0.f && p < 1.f, CCD_EINVAL);
    ccd::DecAttnParams a = ccd::DecAttnParams();
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.out = const_cast<ccd_bf16*>(out); a.ldo = ldo;
    a.lse = const_cast<float*>(lse); a.d_out = d_out;
    a.tokens = (const long long*)tokens; a.key_len = key_len; a.pad_idx = pad_idx; a.causal = causal;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale = scale; a.seed = seed;
    a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    ccd_drop_consts(p, &a.thr, &a.keep_scale);
    const int rc = ccd_dec_attn_check(a);
    if (rc != CCD_OK) return rc;
    CCD_CHECK(lddk % 8 == 0 && lddv % 8 == 0 && CCD_ALIGNED16(dk) && CCD_ALIGNED16(dv) && CCD_ALIGNED16(d_out) && ldo % 8 == 0,
              CCD_ESHAPE);
    if (B == 0) return CCD_OK;
    if (ccd_sattn_mfma(a) && lddq % 8 == 0 && CCD_ALIGNED16(dq) && CCD_ALIGNED16(out)) {
        CCD_LAUNCH(ccd::sattn_bwd_kernel, dim3((B * H + 3) / 4), dim3(256), (size_t)ccd::SA_BWD_SMEM, stream, a);
        return ccd_rt_last_error();
    }
    if (ccd_dec_attn_mfma(a) && lddq % 8 == 0 && CCD_ALIGNED16(dq) && CCD_ALIGNED16(out)) {
        CCD_LAUNCH(ccd::xattn_bwd_kernel, dim3(B * H), dim3(256), (size_t)ccd::XA_BWD_SMEM, stream, a);
        return ccd_rt_last_error();
    }
    CCD_LAUNCH(ccd::dec_attn_bwd_kernel, dim3(B * H), dim3(256), (size_t)ccd::dec_attn_bwd_smem(Tq, Tk), stream, a);
    return ccd_rt_last_error();
}
int ccd_tf_loss_fwd(const float* logits, long ldl, int C, const int64_t* targets, int rows, int T, int pad_idx,
                    float* row_lse, float* acc, void* stream) {
    CCD_CHECK(logits && targets && row_lse && acc && rows >= 0, CCD_EINVAL);
    CCD_CHECK(C > 0 && C <= 128 && ldl >= C && T > 0 && rows % T == 0, CCD_ESHAPE);
    const int rc = ccd_rt_memset_async(acc, 0, 2 * sizeof(float), stream);
    if (rc) return rc;
    if (rows == 0) return CCD_OK;
    const int tf_blocks = (rows + 3) / 4 < 2 * ccd_rt_num_cus() ? (rows + 3) / 4 : 2 * ccd_rt_num_cus();
    CCD_LAUNCH(ccd::tf_loss_fwd_kernel, dim3(tf_blocks), dim3(256), 0, stream, logits, ldl, C, (const long long*)targets,
               rows, T, pad_idx, row_lse, acc);
    return ccd_rt_last_error();
}
int ccd_tf_loss_bwd(const float* logits, long ldl, int C, const int64_t* targets, int rows, int T, int pad_idx,
                    const float* row_lse, const float* acc, const float* upstream, ccd_bf16* d_logits, long ldd, void* stream) {
    CCD_CHECK(logits && targets && row_lse && acc && d_logits && rows >= 0, CCD_EINVAL);
    CCD_CHECK(C > 0 && C <= 128 && ldl >= C && ldd >= C && T > 0 && rows % T == 0, CCD_ESHAPE);
    if (rows == 0) return CCD_OK;
    CCD_LAUNCH(ccd::tf_loss_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, logits, ldl, C, (const long long*)targets,
               rows, T, pad_idx, row_lse, acc, upstream, d_logits, ldd);
    return ccd_rt_last_error();
}
int ccd_greedy_step(const float* logits, long ldl, int C, int B, float* probs, int steps, int step, int64_t* seq,
                    int seq_len, void* stream) {
    CCD_CHECK(logits && probs && seq && B >= 0 && step >= 0 && step < steps, CCD_EINVAL);
    CCD_CHECK(C > 0 && C <= 128 && ldl >= C, CCD_ESHAPE);
    if (B == 0) return CCD_OK;
    CCD_LAUNCH(ccd::greedy_step_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, logits, ldl, C, B, probs, steps, step,
               (long long*)seq, seq_len);
    return ccd_rt_last_error();
}

}  // extern "C"
