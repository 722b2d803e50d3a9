// abi_impl.h - the extern "C" entry points declared in include/ccd_hip.h: argument checks + launch geometry.
// Included by ccd_hip.hip (product, after prelude_hip.h) and by tests/hipsim/ccd_sim.cpp (CPU SIMT executor,
// test infrastructure) - hence launches go through CCD_LAUNCH and the ccd_rt_* helpers of the prelude.
#pragma once

#include "../../include/ccd_hip.h"
#include "kernels/common.h"
#include "kernels/gemm.h"
#include "kernels/layernorm.h"
#include "kernels/attention_fwd.h"
#include "kernels/attention_bwd.h"

#define CCD_CHECK(cond, code) \
    do {                      \
        if (!(cond)) return (code); \
    } while (0)
#define CCD_ALIGNED16(p) ((((uintptr_t)(p)) & 15u) == 0)

template <bool TN>
static int ccd_launch_gemm(const ccd::GemmParams& p, int epilogue, int splits, void* stream) {
    const int tiles = ((p.M + ccd::GEMM_BM - 1) / ccd::GEMM_BM) * ((p.N + ccd::GEMM_BN - 1) / ccd::GEMM_BN);
    const dim3 grid(tiles, 1, splits), block(256);
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

extern "C" {

int ccd_abi_version(void) { return 1; }
const char* ccd_build_info(void) { return "ccd_hip gfx950 bf16-mfma abi1"; }

// ----------------------------------------------------------------------------------------------- GEMM
int ccd_gemm_nt(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, int epilogue, void* C,
                long ldc, void* C2, long ldc2, const float* bias, const float* resid, long ldr,
                const float* rowscale, int rows_per_sample, const ccd_bf16* aux, long ldaux, float alpha,
                int m_fastest, void* stream) {
    CCD_CHECK(A && B && C, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(A) && CCD_ALIGNED16(B) && CCD_ALIGNED16(C), CCD_EINVAL);
    if (M == 0 || N == 0) return CCD_OK;
    CCD_CHECK(M > 0 && N > 0 && K > 0, CCD_EINVAL);
    CCD_CHECK(K % 64 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, CCD_ESHAPE);
    CCD_CHECK(epilogue != CCD_EPI_GELU || (C2 && ldc2 % 8 == 0), CCD_EINVAL);
    CCD_CHECK(epilogue != CCD_EPI_RESID || (resid && ldr % 4 == 0 && rows_per_sample > 0), CCD_EINVAL);
    CCD_CHECK(epilogue != CCD_EPI_DGELU || (aux && ldaux % 8 == 0), CCD_EINVAL);
    ccd::GemmParams p;
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.M = M; p.N = N; p.K = K;
    p.C = C; p.ldc = ldc; p.C2 = C2; p.ldc2 = ldc2; p.bias = bias; p.resid = resid; p.ldr = ldr;
    p.rowscale = rowscale; p.rows_per_sample = rows_per_sample; p.aux = aux; p.ldaux = ldaux;
    p.k_per_split = K; p.m_fastest = m_fastest; p.alpha = alpha;
    return ccd_launch_gemm<false>(p, epilogue, 1, stream);
}

int ccd_gemm_tn(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, int Mc, int epilogue, float* C,
                long ldc, float alpha, int splits, void* stream) {
    CCD_CHECK(A && B && C, CCD_EINVAL);
    CCD_CHECK(CCD_ALIGNED16(A) && CCD_ALIGNED16(B) && CCD_ALIGNED16(C), CCD_EINVAL);
    if (P == 0 || Q == 0 || Mc == 0) return CCD_OK;
    CCD_CHECK(P > 0 && Q > 0 && Mc > 0, CCD_EINVAL);
    CCD_CHECK(P % 8 == 0 && Q % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, CCD_ESHAPE);
    CCD_CHECK(epilogue == CCD_EPI_ATOMIC || epilogue == CCD_EPI_F32, CCD_EINVAL);
    if (splits < 1) {   // pick enough slices to fill the chip (~2 WGs per CU), each a multiple of 64 rows
        const int tiles = ((P + 127) / 128) * ((Q + 127) / 128);
        splits = (2 * ccd_rt_num_cus() + tiles - 1) / tiles;
    }
    int per = (Mc + splits - 1) / splits;
    per = ((per + 63) / 64) * 64;
    splits = (Mc + per - 1) / per;
    CCD_CHECK(epilogue == CCD_EPI_ATOMIC || splits == 1, CCD_EINVAL);
    ccd::GemmParams p;
    p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.M = P; p.N = Q; p.K = Mc;
    p.C = C; p.ldc = ldc; p.C2 = nullptr; p.ldc2 = 0; p.bias = nullptr; p.resid = nullptr; p.ldr = 0;
    p.rowscale = nullptr; p.rows_per_sample = 1; p.aux = nullptr; p.ldaux = 0;
    p.k_per_split = per; p.m_fastest = 0; p.alpha = alpha;
    return ccd_launch_gemm<true>(p, epilogue, splits, stream);
}

// ------------------------------------------------------------------------------------------ LayerNorm
int ccd_ln_fwd(const float* x, const float* gamma, const float* beta, ccd_bf16* y, float* mean, float* rstd, int rows,
               int E, float eps, void* stream) {
    CCD_CHECK(x && gamma && beta && y && mean && rstd, CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && E > 0 && E <= 64 * ccd::LN_MAX_PER_LANE, CCD_ESHAPE);
    CCD_LAUNCH(ccd::ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, rows, E,
               eps);
    return ccd_rt_last_error();
}

int ccd_ln_bwd(const ccd_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* g,
               int accumulate, float* dgamma, float* dbeta, int rows, int E, void* stream) {
    CCD_CHECK(dy && x && mean && rstd && gamma && g && dgamma && dbeta, CCD_EINVAL);
    if (rows == 0) return CCD_OK;
    CCD_CHECK(rows > 0 && E > 0 && E <= 64 * ccd::LN_MAX_PER_LANE, CCD_ESHAPE);
    int blocks = 8 * ccd_rt_num_cus();
    int rpb = (rows + blocks - 1) / blocks;
    rpb = ((rpb + 3) / 4) * 4;
    blocks = (rows + rpb - 1) / rpb;
    if (accumulate)
        CCD_LAUNCH((ccd::ln_bwd_kernel<true>), dim3(blocks), dim3(256), 0, stream, dy, x, mean, rstd, gamma, g, dgamma,
                   dbeta, rows, E, rpb);
    else
        CCD_LAUNCH((ccd::ln_bwd_kernel<false>), dim3(blocks), dim3(256), 0, stream, dy, x, mean, rstd, gamma, g, dgamma,
                   dbeta, rows, E, rpb);
    return ccd_rt_last_error();
}

// ------------------------------------------------------------------------------------------ attention
int ccd_attention_fwd(const ccd_bf16* qkv, ccd_bf16* out, float* lse, int views, int heads, float scale,
                      void* stream) {
    CCD_CHECK(qkv && out && lse, CCD_EINVAL);
    if (views == 0) return CCD_OK;
    CCD_CHECK(views > 0 && heads > 0, CCD_EINVAL);
    CCD_LAUNCH(ccd::attention_fwd_kernel, dim3(views * heads), dim3(256), ccd::ATT_SMEM_BYTES, stream, qkv, out, lse,
               heads, scale);
    return ccd_rt_last_error();
}

int ccd_attention_bwd(const ccd_bf16* qkv, const ccd_bf16* out, const ccd_bf16* d_out, const float* lse,
                      float* delta_ws, ccd_bf16* d_qkv, int views, int heads, float scale, void* stream) {
    CCD_CHECK(qkv && out && d_out && lse && delta_ws && d_qkv, CCD_EINVAL);
    if (views == 0) return CCD_OK;
    CCD_CHECK(views > 0 && heads > 0, CCD_EINVAL);
    CCD_LAUNCH(ccd::attention_bwd_dq_kernel, dim3(views * heads), dim3(512), ccd::ATTB_DQ_SMEM, stream, qkv, out, d_out,
               lse, delta_ws, d_qkv, heads, scale);
    CCD_LAUNCH(ccd::attention_bwd_dkv_kernel, dim3(views * heads), dim3(512), ccd::ATTB_DKV_SMEM, stream, qkv, d_out, lse,
               delta_ws, d_qkv, heads, scale);
    return ccd_rt_last_error();
}

}  // extern "C"
