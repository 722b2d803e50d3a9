/* ccd_hip.h - C ABI of libccd_hip.so: the MI355X (gfx950) kernels of the CCD pretraining step.
 *
 * The reference (TongkunGuan/CCD) is pure Python on stock PyTorch and has no FFI boundary;
 * every entry point below replaces the ATen/cuDNN kernels that one reference call site launches implicitly.
 * The reference line each one stands in for is cited per function (paths relative to /root/reference).
 *
 * Conventions
 *   - plain pointers to DEVICE memory, explicit sizes, `void* stream` = hipStream_t (0 = default stream);
 *   - bf16 tensors are raw uint16 storage, fp32 tensors float; all tensors dense row-major unless a leading
 *     dimension (ld*) is passed, 16-byte aligned;
 *   - returns 0 on success, a negative CCD_E* code for argument errors, a positive hipError_t otherwise;
 *   - never allocates, never synchronises, re-entrant; work is enqueued on `stream`.
 *   - dynamic shapes (number of selected character rows) stay on the device: kernels take `const int* d_rows`
 *     where noted and are launched for the worst case.
 */
#ifndef CCD_HIP_H
#define CCD_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCD_OK 0
#define CCD_EINVAL (-1)   /* bad pointer / size / alignment */
#define CCD_ESHAPE (-2)   /* shape not supported by this kernel (documented per function) */

typedef uint16_t ccd_bf16;

/* library identification; abi version is bumped on any signature change */
int ccd_abi_version(void);
const char* ccd_build_info(void);

/* Kernel-selection policy.  Which tile shape serves a product is decided by a small table of integers that is read ONCE
 * from the environment (CCD_GEMM_256, CCD_GEMM_256_MIN_M, CCD_GEMM_256_MIN_N, CCD_GEMM_256_F32, CCD_GEMM_256_DEEP,
 * CCD_GEMM_ROW384, CCD_ROWGEMM, CCD_LN_BWD_BPC, CCD_DEC_ATTN_SIMT, CCD_ATTN_SKEW, CCD_CU_RESERVE, CCD_LAB) and can be
 * changed at run time by key (lower case, without the prefix) - the launch path never calls getenv.  Unknown key:
 * CCD_EINVAL.  No reference counterpart: the reference leaves kernel selection to ATen / cuDNN heuristics.
 *   rowgemm    0: the row-wise epilogues (ccd_gemm_nt_resid_ln / _lnbwd) on gemm_row384.h's tile (N <= 384 only)
 *              1 (default): LayerNorm-backward product on rowgemm.h (N = 128, 256, 384, 512);
 *                 residual + LayerNorm on gemm_row384.h (N <= 384) / rowgemm.h (N = 512)
 *              2: rowgemm.h for both epilogues wherever its shapes allow (tests)
 *   attn_skew  cycles / 64 by which waves 4..7 of the attention-backward kernels trail waves 0..3 (lab, default 0)
 *   attn_tr    1 (default): dK / dV kernel on double-buffered LDS-DMA row images with transposing LDS reads; 0: four register-staged images
 *   gemm_tn384 1 (default): ViT weight gradients (P % 384 == 0, Q % 192 == 0) on the XCD-grouped kernel of gemm_tn384.h, 0: 128-square
 *              kernel, 2: never as a pair;  gemm_tn384_min_tiles (6): smallest single product it takes;  gemm_tn384_geom 2: also
 *              512x128 tiles for the E = 512 shapes (tested, no gain)
 *   cu_reserve compute units the persistent grids leave free (set while an RCCL gradient reducer is attached)
 *   cu_reserve_window -1 (default): every launch leaves them free; N >= 0: only the next cu_reserve_left launches do - the
 *              gradient reducer sets cu_reserve_left = N whenever it starts a bucket's all-reduce (ccd_amd/parallel.py)
 *   lab        scratch switch of the lab harnesses under tools/ (0 in production) */
int ccd_policy_set(const char* key, int value);
int ccd_policy_get(const char* key, int* value);

/* ---------------------------------------------------------------- GEMM family (nn.Linear / conv-as-GEMM)
 * epilogue codes (ccd_gemm_*: `epilogue`):                                                              */
#define CCD_EPI_BF16 0     /* C(bf16) = acc + bias                                                       */
#define CCD_EPI_GELU 1     /* C(bf16) = u = acc + bias, C2(bf16) = gelu(u)   Mlp.fc1+act, vit.py:59-61    */
#define CCD_EPI_RESID 2    /* C(f32) = resid + (acc + bias) * rowscale[row / rows_per_sample]  Block :109 */
#define CCD_EPI_F32 3      /* C(f32) = acc + bias                                                        */
#define CCD_EPI_ATOMIC 4   /* C(f32) += acc   (split-K partial sums; NT: K >= 16384 is cut into slices of 8192)   */
#define CCD_EPI_DGELU 5    /* C(bf16) = acc * gelu'(aux); C2 (optional, bf16) = gelu(aux)                */

/* C[M,N] = A[M,K] . B[N,K]^T   (F.linear(x, W): Dino/modules/vision_transformer.py:60,63,82,90,325-327)
 * K % 64 == 0, N % 8 == 0.  m_fastest: tile order hint (1 when B is the large operand). */
int ccd_gemm_nt(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, int epilogue,
                void* C, long ldc, void* C2, long ldc2, const float* bias, const float* resid, long ldr,
                const float* rowscale, int rows_per_sample, const ccd_bf16* aux, long ldaux, float alpha,
                int m_fastest, const int* d_rows, int rows_mul, float* colsum, void* stream);
/* Residual product with the NEXT LayerNorm folded into its epilogue (Block.forward, vision_transformer.py:107-113:
 * x = x + drop_path(f(...)); the following norm1 / norm2 / norm of the stream):
 *   C[M,N] (f32) = resid + (A . B^T + bias) * rowscale[row / rows_per_sample]
 *   y[M,N] (bf16) = (C - mean) * rstd * ln_gamma + ln_beta,  mean / rstd [M] saved for ccd_ln_bwd
 * One workgroup owns whole rows: N in {128, 256, 384, 512} with K % 128 == 0 (K % 192 at N = 384) runs the row-owner
 * kernel (rowgemm.h), any other N <= 384 (N % 8 == 0, K % 64 == 0) the 128 x 384 tile of gemm_row384.h;
 * replaces ccd_gemm_nt(EPI_RESID) + ccd_ln_fwd. */
int ccd_gemm_nt_resid_ln(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, float* C, long ldc,
                         const float* bias, const float* resid, long ldr, const float* rowscale, int rows_per_sample,
                         const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* y, long ldy, float* mean,
                         float* rstd, void* stream);
/* A data-gradient product whose result IS the gradient of a LayerNorm output, with that LayerNorm's backward pass in the
 * epilogue (autograd of nn.LayerNorm(eps=1e-6) in Block.forward, vision_transformer.py:99,103,107-113):
 *   dy = A . B^T  (never written);  xhat = (x - mean) * rstd;  dx = rstd * (dy*gamma - mean_row(dy*gamma) - xhat * mean_row(dy*gamma*xhat))
 *   g (f32) = (accumulate ? g : 0) + dx;  dgamma += colsum(dy * xhat);  dbeta += colsum(dy)
 *   optional tail: gb (bf16) = g_new * rowscale[row / rows_per_sample], dbias += colsum(gb)
 * Same shape rules as ccd_gemm_nt_resid_ln.  Replaces ccd_gemm_nt(EPI_BF16) + ccd_ln_bwd. */
int ccd_gemm_nt_lnbwd(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, const float* x, long ldx,
                      const float* mean, const float* rstd, const float* gamma, float* g, long ldg, int accumulate,
                      float* dgamma, float* dbeta, ccd_bf16* gb, long ldgb, const float* rowscale, int rows_per_sample,
                      float* dbias, void* stream);
/* The same with the residual-gradient stream g in bf16 (ABI 10; N in {128, 256, 384} - CCD_ESHAPE otherwise): g is read as bf16,
 * accumulated in fp32 and rounded once per writer - half the bytes of the stream that every LayerNorm backward of a transformer
 * block reads and rewrites (the reference keeps that gradient in fp32 under autocast; gated by the parity tests). */
int ccd_gemm_nt_lnbwd_g16(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, const float* x, long ldx,
                          const float* mean, const float* rstd, const float* gamma, ccd_bf16* g, long ldg, int accumulate,
                          float* dgamma, float* dbeta, ccd_bf16* gb, long ldgb, const float* rowscale, int rows_per_sample,
                          float* dbias, void* stream);
/* The same with a SECOND LayerNorm of the same rows folded into the epilogue (ABI 11; N == 384 only - CCD_ESHAPE otherwise): a
 * segmentation tap (Dino/modules/vision_transformer.py:245-249: norm_seg[j](x) of the rows that norm1 of the next block also
 * normalises - same mean / rstd, other gamma).  LayerNorm's backward is linear in dy * gamma, so
 *   dx = core(dy * gamma + tap_dy * tap_gamma);  tap_dgamma += colsum(tap_dy * xhat);  tap_dbeta += colsum(tap_dy)
 * and x, g, gb move once instead of twice.  Replaces ccd_gemm_nt_lnbwd_g16 + ccd_ln_bwd of the tap (tap_dy: [M, N] bf16). */
int ccd_gemm_nt_lnbwd_tap_g16(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, const float* x, long ldx,
                              const float* mean, const float* rstd, const float* gamma, ccd_bf16* g, long ldg, int accumulate,
                              float* dgamma, float* dbeta, ccd_bf16* gb, long ldgb, const float* rowscale, int rows_per_sample,
                              float* dbias, const ccd_bf16* tap_dy, long ld_tap, const float* tap_gamma, float* tap_dgamma,
                              float* tap_dbeta, void* stream);
/* The whole MLP branch of a transformer block in one launch (Mlp.forward + the residual of Block.forward,
 * Dino/modules/vision_transformer.py:59-65,107-113, and the LayerNorm that consumes the stream next, :99/:156):
 *   h = gelu(bf16(y . W1^T + b1)) ;  out (f32) = resid + (h . W2^T + b2) * rowscale[row / rows_per_sample]
 *   ln_y (bf16) = (out - mean) * rstd * ln_gamma + ln_beta ;  ln_mean / ln_rstd [M] saved for ccd_ln_bwd
 * The [M, H] hidden activation stays on chip; `u` (optional, [M, H] bf16) receives the pre-activation the backward
 * pass needs, `gact` (optional, only with u) gelu(u) - the operand of the weight-gradient product dW2 = gb^T . gelu(u): stored
 * here, the backward's gelu'(u) product (CCD_EPI_DGELU) neither looks Phi up a second time nor writes gelu(u).
 * W1 = fc1.weight [H, E], W2 = fc2.weight [E, H] (bf16).  E in {128, 256, 384, 512}, H % 64 == 0.
 * Replaces ccd_gemm_nt(EPI_GELU) + ccd_gemm_nt_resid_ln. */
int ccd_mlp_fused(const ccd_bf16* y, long ldy, const ccd_bf16* w1, long ld1, const float* b1, const ccd_bf16* w2, long ld2,
                  const float* b2, const float* resid, long ldr, const float* rowscale, int rows_per_sample, float* out,
                  long ldc, const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* ln_y, long ld_y,
                  float* ln_mean, float* ln_rstd, ccd_bf16* u, long ldu, ccd_bf16* gact, long ldga, int M, int E, int H, void* stream);
/* The whole second half of a transformer block in one launch (round 5): the tail of the attention branch in front of the fused MLP,
 *     x_mid = resid + rowscale1[row / rps] * (a . Wp^T + bp)            vision_transformer.py:91 (proj), :108 (residual + DropPath)
 *     y2    = LayerNorm(x_mid) * ln2_gamma + ln2_beta                   :109 (norm2)
 *     out   = x_mid + rowscale2[row / rps] * (gelu(y2 . W1^T + b1) . W2^T + b2);   ln_y = LayerNorm(out) * ln_gamma + ln_beta
 * x_mid never leaves the accumulator registers and y2 never leaves the operand registers: a forward pass that keeps nothing
 * (xmid = y2 = mean2 = rstd2 = u = NULL: the teacher) reads a and resid and writes out and ln_y - 604 MB per 131 072 rows instead
 * of the 1 208 MB of ccd_gemm_nt_resid_ln + ccd_mlp_fused.  With xmid / y2 / mean2 / rstd2 / u (all five or none) the tensors
 * the backward pass needs are written on the way.  E in {128, 256, 384}; with a DropPath scale rows_per_sample % 128 == 0
 * (CCD_ESHAPE otherwise: the caller takes the two separate launches); rowscale2 needs xmid (a dropped MLP branch still runs its
 * products - no branch around the weight ring - and reads x_mid back: out == x_mid exactly, u holds the real pre-activation).
 * tap_y (optional, with tap_gamma / tap_beta): a second LayerNorm of the same output rows, LayerNorm(out) * tap_gamma + tap_beta - the
 * segmentation tap that follows some blocks (vision_transformer.py:245-249) - instead of a ccd_ln_fwd pass over out.
 * Replaces ccd_gemm_nt_resid_ln + ccd_mlp_fused (+ ccd_ln_fwd). */
int ccd_proj_mlp_fused(const ccd_bf16* a, long lda, const ccd_bf16* wp, long ldp, const float* bp, const float* resid, long ldr,
                       const float* rowscale1, const float* ln2_gamma, const float* ln2_beta, float* xmid, long ldxm, ccd_bf16* y2,
                       long ldy2, float* mean2, float* rstd2, const ccd_bf16* w1, long ld1, const float* b1, const ccd_bf16* w2,
                       long ld2, const float* b2, const float* rowscale2, int rows_per_sample, float* out, long ldc,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* ln_y, long ld_y, float* ln_mean,
                       float* ln_rstd, ccd_bf16* u, long ldu, const float* tap_gamma, const float* tap_beta, ccd_bf16* tap_y, long ld_tap,
                       int M, int E, int H, void* stream);
/* The same, and gelu(u) [M, H] bf16 is stored too (ABI 12; u required): the packed second-product operands of the forward kernel ARE that
 * tensor, so the backward pass neither gathers Phi a second time nor writes gelu(u) - what ccd_mlp_bwd_fused expects of its caller. */
int ccd_proj_mlp_fused_gact(const ccd_bf16* a, long lda, const ccd_bf16* wp, long ldp, const float* bp, const float* resid, long ldr,
                       const float* rowscale1, const float* ln2_gamma, const float* ln2_beta, float* xmid, long ldxm, ccd_bf16* y2,
                       long ldy2, float* mean2, float* rstd2, const ccd_bf16* w1, long ld1, const float* b1, const ccd_bf16* w2,
                       long ld2, const float* b2, const float* rowscale2, int rows_per_sample, float* out, long ldc,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, ccd_bf16* ln_y, long ld_y, float* ln_mean,
                       float* ln_rstd, ccd_bf16* u, long ldu, ccd_bf16* gact, long ldga, const float* tap_gamma, const float* tap_beta, ccd_bf16* tap_y, long ld_tap,
                       int M, int E, int H, void* stream);
/* The data-gradient chain of the MLP branch in one launch - the backward of the block half above (Mlp.forward + LayerNorm-2,
 * Dino/modules/vision_transformer.py:59-65,110 in autograd order; ABI 12):
 *   dh = gb . W2 ;  du = dh * gelu'(u) ;  dy2 = du . W1 ;  LayerNorm-2 backward of dy2 as in ccd_gemm_nt_lnbwd_g16
 *   (g (+)= dx on the bf16 stream, dgamma / dbeta +=, gb_out = bf16(g * rowscale[row / rows_per_sample]), dbias += colsum(gb_out))
 * w2t = fc2.weight^T [H, E], w1t = fc1.weight^T [E, H] (the transposed bf16 mirrors), u [M, H] the stored pre-activation.  du [M, H] is
 * written ONCE, for the weight-gradient pair that follows (ccd_gemm_tn_pair: {gb, gelu(u)}, {du, y2}), and never read back; db1 +=
 * column sums of du (fc1.bias gradient).  gb_out must not alias gb.  E in {256, 384}, H % 128 == 0; CCD_ESHAPE otherwise: the caller
 * takes ccd_gemm_nt(EPI_DGELU) + ccd_gemm_nt_lnbwd_g16.  Replaces those two. */
int ccd_mlp_bwd_fused(const ccd_bf16* gb, long ldgb, const ccd_bf16* w2t, long ld2, const ccd_bf16* w1t, long ld1, const ccd_bf16* u,
                      long ldu, ccd_bf16* du, long lddu, float* db1, const float* x, long ldx, const float* mean, const float* rstd,
                      const float* gamma, ccd_bf16* g, long ldg, int accumulate, float* dgamma, float* dbeta, ccd_bf16* gb_out, long ld_gbo,
                      const float* rowscale, int rows_per_sample, float* dbias, int M, int E, int H, void* stream);

/* colsum (optional, epilogues BF16 / DGELU): [N] fp32, += column sums of the output (bias gradient of the producer).
 * CCD_EPI_GELU accepts C == NULL (only gelu(u) is stored: forward passes that keep no activations). */
/* C[P,Q] (+)= sum_m A[m,P] * B[m,Q]   (weight gradients dW = dY^T X of every Linear; autograd of the above)
 * P % 8 == 0, Q % 8 == 0; epilogue CCD_EPI_ATOMIC accumulates into fp32 C (split over m), CCD_EPI_F32 stores. */
int ccd_gemm_tn(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, int Mc, int epilogue,
                float* C, long ldc, float alpha, int splits, const int* d_rows, int rows_mul, void* stream);
/* The same weight-gradient product with the BIAS gradient of the same Linear in one pass over dY:
 * C[P,Q] += A^T . B (fp32 atomics, split over m) and colsum_a[P] += column sums of A - the rows of dY are summed while they
 * pass through the loader's registers (replaces ccd_gemm_tn + ccd_colsum_bf16: one read of dY less). */
int ccd_gemm_tn_colsum(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, int Mc, float* C, long ldc,
                       float* colsum_a, int splits, void* stream);
/* Two weight-gradient products over the SAME contraction rows in one launch: C1[P1,Q1] += A1^T . B1 and C2[P2,Q2] += A2^T . B2
 * (fp32 atomics) - the pairs autograd produces together: {fc2.weight, fc1.weight} once gelu'(u) is applied, {proj.weight,
 * qkv.weight} once the attention backward has run (Dino/modules/svtr.py:92-145).  With P % 384 == 0, Q % 192 == 0 and
 * Mc % 32 == 0 both run in gemm_tn384.h's XCD-grouped launch and share ONE atomic epilogue per workgroup; any other shape
 * falls back to two ccd_gemm_tn calls.  Results are those of the two separate calls up to fp32 summation order. */
int ccd_gemm_tn_pair(const ccd_bf16* A1, long lda1, const ccd_bf16* B1, long ldb1, int P1, int Q1, float* C1, long ldc1,
                     const ccd_bf16* A2, long lda2, const ccd_bf16* B2, long ldb2, int P2, int Q2, float* C2, long ldc2, int Mc,
                     void* stream);
/* The same with a caller-owned fp32 WORKSPACE (round 4): each contraction slice stores its partial tile into its own plane of ws
 * (plain stores) and one reduction pass adds the planes to C1 / C2 - replaces 75 MB of fp32 atomics per MLP pair.  ws_floats >=
 * ccd_gemm_tn_pair_ws_floats(P1, Q1, P2, Q2) guarantees the path is taken (0 = shapes the grouped kernel does not cover); a null or
 * too small workspace falls back to ccd_gemm_tn_pair's atomics.  The workspace is scratch: its contents are undefined afterwards and
 * launches that share it must be ordered on one stream. */
long ccd_gemm_tn_pair_ws_floats(int P1, int Q1, int P2, int Q2);
int ccd_gemm_tn_pair_ws(const ccd_bf16* A1, long lda1, const ccd_bf16* B1, long ldb1, int P1, int Q1, float* C1, long ldc1,
                        const ccd_bf16* A2, long lda2, const ccd_bf16* B2, long ldb2, int P2, int Q2, float* C2, long ldc2, int Mc,
                        float* ws, long ws_floats, void* stream);
/* d_rows (optional, both GEMMs): device int; the effective row count (NT: M, TN: Mc) is
 * min(static value, d_rows[0] * rows_mul) so data-dependent row counts never reach the host. */

/* ---------------------------------------------------------------- LayerNorm (eps 1e-6), vit.py:99,103,156,162-166 */
int ccd_ln_fwd(const float* x, const float* gamma, const float* beta, ccd_bf16* y, float* mean, float* rstd,
               int rows, int E, float eps, void* stream);
/* g (+)= dx ; dgamma += ..., dbeta += ... (fp32 atomics; caller zeroes them once per step).
 * Optional fused tail: gb(bf16) = g_new * rowscale[row / rows_per_sample] (the gradient entering the next residual
 * branch, DropPath scale applied) and dbias += column sums of gb. */
int ccd_ln_bwd(const ccd_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
               float* g, int accumulate, float* dgamma, float* dbeta, ccd_bf16* gb, const float* rowscale,
               int rows_per_sample, float* dbias, int rows, int E, void* stream);

int ccd_ln_bwd_g16(const ccd_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                   ccd_bf16* g /* bf16 stream, see ccd_gemm_nt_lnbwd_g16 */, int accumulate, float* dgamma, float* dbeta, ccd_bf16* gb,
                   const float* rowscale, int rows_per_sample, float* dbias, int rows, int E, void* stream);

/* ---------------------------------------------------------------- attention, vit.py:80-92 (T = 256, head_dim = 64)
 * qkv [views, 256, 3, heads, 64] bf16 (the layout Attention.forward reshapes to), out [views, 256, heads*64]  */
int ccd_attention_fwd(const ccd_bf16* qkv, ccd_bf16* out, float* lse, int views, int heads, float scale,
                      void* stream);
/* d_qkv_bias (nullable, fp32 [3 * heads * 64], ACCUMULATED): the gradient of the qkv bias (vit.py:75 `qkv_bias`; what autograd's
 * Linear backward computes as grad_output.sum(0) over d_qkv) without a pass over d_qkv:
 *   q part  column sums of the fp32 dQ tiles, inside the dQ kernel; needs bias_ws = ccd_attention_bwd_ws_floats(views, heads)
 *           floats of scratch (per-workgroup partial rows, plain stores)
 *   k part  identically zero (the softmax is shift invariant): nothing is added
 *   v part  = column sums of d_out (every softmax row sums to 1), which the caller supplies WITHOUT touching d_out:
 *           dout_colsum_mat == null: dout_colsum_vec [E] IS colsum(d_out);
 *           otherwise colsum(d_out) = dout_colsum_vec [E] . dout_colsum_mat [E, E] (row-major, ld_mat) - for d_out = gb . Wproj
 *           (vit.py:90-91) that is (proj.bias gradient) . (proj.weight).
 * A third, tiny launch adds the partial rows up and forms the matvec. */
long ccd_attention_bwd_ws_floats(int views, int heads);
int ccd_attention_bwd(const ccd_bf16* qkv, const ccd_bf16* out, const ccd_bf16* d_out, const float* lse,
                      float* delta_ws, ccd_bf16* d_qkv, int views, int heads, float scale, float* d_qkv_bias, float* bias_ws,
                      const float* dout_colsum_vec, const float* dout_colsum_mat, long ld_mat, void* stream);

/* ---------------------------------------------------------------- patch embedding, vit.py:128-131,225-236
 * img [views,3,32,128] fp32 NCHW; w [E,3,4,4]; pos [256,E] = the bicubically resampled pos_embed; out fp32 [views*256,E] */
int ccd_patch_embed_fwd(const float* img, const float* w, const float* bias, const float* pos, float* out, int views,
                        int E, void* stream);
/* g = d(out) fp32 [views*256,E]; d_w [E,48], d_bias [E], d_pos [256,E] are ACCUMULATED (fp32 atomics).
 * Caller-provided workspaces: ws_g bf16 [views*256, E] (receives bf16(g)), ws_patches bf16 [views*256, 48].
 * d_w = bf16(g)^T . bf16(patches) on the MFMA path (TN GEMM), d_bias = column sums of bf16(g). */
int ccd_patch_embed_bwd(const float* img, const float* g, float* d_w, float* d_bias, float* d_pos, ccd_bf16* ws_g,
                        ccd_bf16* ws_patches, int views, int E, void* stream);
/* the same on a bf16 stream g: it is the TN product's operand as it lies (no ws_g) */
int ccd_patch_embed_bwd_g16(const float* img, const ccd_bf16* g, float* d_w, float* d_bias, float* d_pos, ccd_bf16* ws_patches, int views,
                            int E, void* stream);
/* C[M,N] (+)= op(A) . B, fp32, tiny problems (bicubic pos-embed resampling = fixed 256x256 linear map, vit.py:182-201) */
int ccd_small_matmul_f32(const float* A, const float* B, float* C, int M, int N, int K, int trans_a, int accumulate,
                         void* stream);
/* out[n] += sum_rows x[row,n] (bias gradients) */
int ccd_colsum_bf16(const ccd_bf16* x, long ld, int rows, int N, const int* d_rows, int rows_mul, float* out,
                    void* stream);
/* bf16 mirrors (and transposed mirrors) of a batch of fp32 matrices */
typedef struct ccd_mirror_desc {
    const float* src;
    ccd_bf16* dst;    /* [rows, cols] or NULL */
    ccd_bf16* dst_t;  /* [cols, rows] or NULL */
    int rows, cols;
    int tile_begin;   /* first 64x64 tile of this matrix in the launch (prefix sum over ceil(rows/64) * ceil(cols/64); ABI 9: was 32x32) */
    int pad_;
} ccd_mirror_desc;
int ccd_mirror_bf16(const ccd_mirror_desc* d_descs, int ndesc, int total_tiles, void* stream);
int ccd_cast_bf16(const float* src, ccd_bf16* dst, long n, void* stream);
/* dst(bf16)[r,:] = src(f32)[r,:] * rowscale[r / rows_per_sample] (rowscale may be NULL): residual gradient -> branch */
int ccd_scale_cast_rows(const float* src, ccd_bf16* dst, const float* rowscale, int rows_per_sample, long rows, int E,
                        void* stream);

/* ---------------------------------------------------------------- character-region path (id maps: uint8, 255 = none)
 * label_cluster.forward, Dino/utils/DBSCAN.py:65-103: mask [images,32,128] (nonzero = text) -> idmap            */
int ccd_ccl_label(const float* mask, uint8_t* idmap, int images, void* stream);
int ccd_mask_to_idmap(const float* mask, uint8_t* idmap, int images, void* stream);
/* softmax(seg)[:,1] > 0.5 of the first `images` images (dino_vision.py:65-66); seg [>=images,2,32,128] fp32 */
int ccd_seg_to_mask(const float* seg_logits, float* mask, int images, void* stream);
/* ---------------------------------------------------------------- data pipeline
 * clusterpixels(im, 2), mask_create/generate_mask.py:13-29 (= Dino/utils/kmeans.py:7-23): 2-means of the gray values of a
 * word image + "the cluster that owns the border is background".  A RAGGED batch: image i is gray[offsets[i] ..
 * offsets[i+1]) with hw[2i] rows x hw[2i+1] columns (device arrays; offsets has images + 1 entries); mask gets 0 / 1
 * at the same offsets.  Integer / fp64 arithmetic with a fixed operation order: bit-exact against oracle/datapipe_np.py. */
int ccd_kmeans2_mask(const uint8_t* gray, const long* offsets, const int* hw, uint8_t* mask, int images, void* stream);
/* The three views of ImageDatasetSelfSupervisedKmeans._process_training (datasetsupervised_kmeans.py:48-87) from resized
 * uint8 images img [B,H,W,3]: out fp32 [B,3,3,H,W] = (plain, augment(params[b,0]), warp_theta(augment(params[b,1]))),
 * each normalised with mean3 / std3 (HOST arrays of 3 floats; dataset.py:79-80).  params fp32 [B,2,96] (layout in
 * kernels/datapipe.h), theta fp32 [B,3,3] = the `metrics` tensor the model receives (identity = no warp).
 * Two launches.  (1) The reference's imgaug chain (augmentation_pipelines.py:120-205): one workgroup per (sample, view) holds
 * the uint8 image in LDS and applies ONE member of each group in the reference's order - `arithmetic` (noises, dropouts,
 * JpegCompression, Emboss / EdgeDetect / DirectedEdgeDetect, the PIL filter presets ...), `color` (cv2's 8-bit HSV members,
 * Grayscale, quantisation ...), `Blur` (Gaussian / average / median / motion / bilateral blur, Sharpen), `contrast` (gamma,
 * linear, sigmoid, log, per-channel histogram equalisation) - rounding to uint8 between the groups like imgaug, and stages the
 * result in staged_ws [B,2,H,W,3] (caller-allocated scratch).  (2) Normalisation and the affine warp of view 2 read those.
 * H * W is bounded by the chain's LDS images (CCD_ESHAPE beyond).
 * overlay (optional): fp16 [overlay_layers, 2, H, W] = (alpha, intensity) planes of imgaug's cloud layers (`weather` group: Fog,
 * Clouds - augmentation_pipelines.py:192-193); a parameter row with params[81] = n > 0 blends its layers params[82] .. + n - 1 after
 * the contrast group: out = trunc(clip((1 - alpha) * v + alpha * intensity, 0, 255)).  The planes are drawn on the host
 * (ccd_amd/dataset/weather.py: frequency noise by FFT, a few hundred microseconds per layer).
 * warp_maps (optional): fp32 [warp_count, 2, H, W] = the SOURCE position (x, y), in pixels, of every output pixel of a
 * piecewise-affine warp (imgaug PiecewiseAffine, the finetuning geometry's second member, dataset_pretrain.py:156; drawn on the
 * host); a sample whose view-2 row has params[84] = m > 0 samples map m - 1 (bilinear, zero fill) instead of theta. */
int ccd_augment_views(const uint8_t* img, const float* params, const float* theta, float* out, uint8_t* staged_ws, int batch,
                      int height, int width, const float* mean3, const float* std3, const uint16_t* overlay, int overlay_layers,
                      const float* warp_maps, int warp_count, void* stream);
/* affine_grid(theta[:, :2]) + grid_sample(bilinear) > 0.1, dino_vision.py:72-77 / train.py:234-236; theta row stride in floats */
int ccd_warp_idmap(const uint8_t* src, const float* theta, int theta_stride, uint8_t* dst, int images, void* stream);
/* ABIDINOModel.attention, dino_vision.py:38-49, in sparse form: per token up to 4 (plane, normalised weight) pairs -
 * tok_plane [views,256,4] (255 = unused slot), tok_coef [views,256,4]; the central 2x2 of a 4x4 cell touches one plane in
 * the labelled view, possibly several in the warped view (components one pixel apart can end up side by side) */
int ccd_region_stats(const uint8_t* idmap, uint8_t* tok_plane, float* tok_coef, uint8_t* present, int views,
                     void* stream);
/* dino_vision.py:82-85: nsel[b], offset[b], total[0] = M, new_index [batch,26] */
int ccd_select_scan(const uint8_t* present, int batch, int* nsel, int* offset, int* total, uint8_t* new_index,
                    void* stream);
/* rows [2M,E] bf16 = gathered pooled character vectors of both views (dino_vision.py:44-47,87); views = 2*batch */
int ccd_region_pool_fwd(const ccd_bf16* feat, const uint8_t* tok_plane, const float* tok_coef, const int* nsel,
                        const int* offset, const int* total, ccd_bf16* rows, int batch, int E, void* stream);
int ccd_region_pool_bwd(const ccd_bf16* d_rows, const uint8_t* tok_plane, const float* tok_coef, const int* nsel,
                        const int* offset, const int* total, ccd_bf16* d_feat, int batch, int E, void* stream);
int ccd_idmap_to_planes(const uint8_t* idmap, float* planes, int images, void* stream);
int ccd_planes_to_idmap(const float* planes, uint8_t* idmap, int images, void* stream);

/* ---------------------------------------------------------------- DINOHead pieces, vit.py:313,326 */
int ccd_l2norm_fwd(const ccd_bf16* x, ccd_bf16* y, float* inv, int max_rows, const int* d_rows, int rows_mul, int D,
                   void* stream);
int ccd_l2norm_bwd(const ccd_bf16* x, const float* inv, const ccd_bf16* dy, ccd_bf16* dx, int max_rows,
                   const int* d_rows, int rows_mul, int D, void* stream);
int ccd_weightnorm_fwd(const float* v, const float* g, ccd_bf16* w, ccd_bf16* w_t, float* inv, int K, int D,
                       void* stream);
int ccd_weightnorm_bwd(const float* v, const float* g, const float* inv, const float* dw, float* dv, float* dg, int K,
                       int D, void* stream);

/* ---------------------------------------------------------------- DINOLoss, Dino/loss/Dino_loss.py:59-143
 * student/teacher logits fp32 [>=2M, K]; d_m = device M; stats [max_rows,4]; loss_out accumulates the scalar */
int ccd_dino_loss_fwd(const float* s_logits, const float* t_logits, const float* center, int K, const int* d_m,
                      int max_rows, float student_temp, float teacher_temp, float* stats, float* loss_out,
                      void* stream);
int ccd_dino_loss_bwd(const float* s_logits, const float* t_logits, const float* center, int K, const int* d_m,
                      int max_rows, float student_temp, float teacher_temp, const float* stats, float grad_scale,
                      const float* d_grad_scale /* optional device scalar, multiplied in */, ccd_bf16* d_logits,
                      void* stream);
/* The head's last layer AND the distillation loss in one pass, the [2M, K] logits never written (ABI 10; replaces the chain
 * ccd_gemm_nt(EPI_F32) x 2 -> ccd_dino_loss_fwd / _bwd of vision_transformer.py:326-327 + Dino_loss.py:81-105 where D == 256 and
 * K % 512 == 0): logits s = zs . ws^T, t = zt . wt^T are formed tile by tile in registers and folded into per-row online-softmax
 * state; the backward entry recomputes the same tiles and writes the bf16 logit gradient d_logits [>= 2M, K] that the head's
 * weight / data gradient products read.  zs / zt [max_rows, 256] bf16 (L2-normalised rows), ws / wt [K, 256] bf16 (weight-normed
 * last layers), center [K] fp32, d_m = device M, ws_part = workspace of ccd_head_loss_ws_floats(max_rows, K) floats,
 * stats [max_rows, 4] (written by _fwd, read by _bwd; base-2 domain: m_s, 1 / l_s, m_t, 1 / l_t), loss_out accumulates the scalar.
 * CCD_ESHAPE for other shapes: the caller takes the unfused chain. */
long ccd_head_loss_ws_floats(int max_rows, int K);
int ccd_head_loss_fwd(const ccd_bf16* zs, long ld_zs, const ccd_bf16* zt, long ld_zt, const ccd_bf16* ws, long ld_ws,
                      const ccd_bf16* wt, long ld_wt, const float* center, int K, int D, const int* d_m, int max_rows,
                      float student_temp, float teacher_temp, float* ws_part, float* stats, float* loss_out, void* stream);
int ccd_head_loss_bwd(const ccd_bf16* zs, long ld_zs, const ccd_bf16* zt, long ld_zt, const ccd_bf16* ws, long ld_ws,
                      const ccd_bf16* wt, long ld_wt, const float* center, int K, int D, const int* d_m, int max_rows,
                      float student_temp, float teacher_temp, const float* stats, float grad_scale,
                      const float* d_grad_scale /* optional device scalar, multiplied in */, ccd_bf16* d_logits, long ld_d,
                      void* stream);
int ccd_colsum_f32(const float* x, int K, const int* d_rows, int rows_mul, int max_rows, float* out, void* stream);
/* out[k] += sum_d w[k, d] * v[d]  (w [K, D] bf16, D % 256 == 0, v and out fp32).  The teacher centre of Dino_loss.py:133-143 without a
 * pass over the [2M, K] teacher logits: their column sums are (sum of the rows of zn) . W^T - ccd_colsum_bf16 of the l2-normalised
 * bottleneck rows, then this product against the weight-normed last layer (33 MB read instead of 865). */
int ccd_matvec_bf16(const ccd_bf16* w, long ldw, const float* v, int K, int D, float* out, void* stream);
int ccd_center_ema(float* center, const float* batch_sum, int K, const int* d_m, int world, float momentum,
                   void* stream);
/* softmax -> cross_entropy (double softmax) of the seg logits [2*half,2,32,128]; d_logits may be NULL */
int ccd_seg_loss(const float* logits, const float* mask_a, const uint8_t* idmap_b, int half, float grad_scale,
                 float* loss_out, float* d_logits, void* stream);

/* ---------------------------------------------------------------- optimiser, train.py:244-272 */
typedef struct ccd_seg_hyper { float lr_wd, step_size, inv_sqrt_bc2, active; } ccd_seg_hyper;
int ccd_seg_sumsq(const float* grad, const int* chunk_seg, const long* chunk_begin, const int* chunk_len, int nchunks,
                  float* norm2, void* stream);
int ccd_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, ccd_bf16* mirror,
              const int* chunk_seg, const long* chunk_begin, const int* chunk_len, int nchunks,
              const ccd_seg_hyper* hyper, const float* norm2, float clip, float beta1, float beta2, float eps,
              void* stream);
int ccd_clip_scale(float* grad, const int* chunk_seg, const long* chunk_begin, const int* chunk_len, int nchunks,
                   const float* norm2, float clip, void* stream);
/* d_m (optional, device, 2 floats {m, 1 - m}): read at run time instead of the two launch arguments (HIP-graph replays of the step) */
int ccd_ema(float* teacher, const float* student, ccd_bf16* mirror, long n, float m, float one_minus_m, const float* d_m,
            void* stream);

/* ---- segmentation head (Dino/modules/segmentor.py:38-95): channels-last bf16 activations [pixels, C] ----------- */
/* Gather description of an implicit-GEMM convolution: output row r = (n, oy, ox) on a 2^g_h_log2 x 2^g_w_log2 grid;
 * contraction index k = tap*cin + c reads source pixel (oy*s_mul + dy[tap], ox*s_mul + dx[tap]) of an s_h x s_w image
 * (zero outside).  Covers Conv2d 3x3/1x1 forward (segmentor.py:42-45), its data gradient (flipped taps), one output
 * parity class of ConvTranspose2d(4,2,1) (segmentor.py:82,85; c_map scatters row (n,oy,ox) to (n,2oy+c_py,2ox+c_px))
 * and that layer's data gradient (s_mul = 2, 16 taps). */
typedef struct ccd_conv_desc {
    int g_h_log2, g_w_log2, s_h, s_w, s_mul, cin, ntaps;
    signed char dy[16], dx[16];
    int c_map, c_py, c_px;
} ccd_conv_desc;
/* C[rows(->c_map), N] (bf16) = gather(src)[M, ntaps*cin] . W[N, ntaps*cin]^T + bias; optional fp32 column sums /
 * sums of squares of the output (+=) = the BatchNorm batch statistics (replaces F.conv2d / F.conv_transpose2d +
 * the statistics pass of F.batch_norm). */
int ccd_conv_gemm(const ccd_bf16* src, long src_ld, const ccd_conv_desc* desc, const ccd_bf16* W, long ldw, int M, int N,
                  ccd_bf16* C, long ldc, const float* bias, float* colsum, float* colsumsq, void* stream);
/* out[P, ntaps*cin] (fp32) += A[rows, P]^T . gather(src)[rows, ntaps*cin]: the weight gradient of the convolution
 * `desc` describes, contracted over pixels and split over them (fp32 atomics); the patch matrix is never materialised.
 * Conv2d: A = dY, src = layer input.  ConvTranspose2d: A = layer input, src = dY (desc of the data gradient). */
int ccd_conv_wgrad(const ccd_bf16* A, long lda, int P, const ccd_bf16* src, long src_ld, const ccd_conv_desc* desc,
                   long rows, float* out, long ldo, void* stream);
/* cols[rows, ntaps*cin] = gather(src): explicit patch matrix for the weight gradients (TN GEMM operand). */
int ccd_im2col(const ccd_bf16* src, long src_ld, const ccd_conv_desc* desc, long rows, ccd_bf16* cols, void* stream);
/* Train-mode BatchNorm2d + ReLU (segmentor.py:43-44,83-84).  stats = [sum x | sum x^2] over `count` pixels (reduced
 * over ranks by the caller for SyncBatchNorm); finalize writes mean_rstd = [mean | 1/sqrt(var+eps)] and updates the
 * running statistics with torch's rule (unbiased variance). */
int ccd_bn_finalize(const float* stats, float count, float eps, float momentum, float* mean_rstd, float* running_mean,
                    float* running_var, int C, void* stream);
int ccd_bn_relu_fwd(const ccd_bf16* x, long ldx, const float* mean_rstd, const float* gamma, const float* beta,
                    ccd_bf16* y, long ldy, long rows, int C, void* stream);
/* red[0:C] += sum dy*[y>0], red[C:2C] += sum dy*[y>0]*xhat */
int ccd_bn_relu_bwd_reduce(const ccd_bf16* dy, long lddy, const ccd_bf16* x, long ldx, const float* mean_rstd,
                           const float* gamma, const float* beta, float* red, long rows, int C, void* stream);
/* dx = gamma*rstd*(dy*[y>0] - red0/count - xhat*red1/count); dgamma += red_local[C:2C], dbeta += red_local[0:C] */
int ccd_bn_relu_bwd_apply(const ccd_bf16* dy, long lddy, const ccd_bf16* x, long ldx, const float* mean_rstd,
                          const float* gamma, const float* beta, const float* red, float count, const float* red_local,
                          float* dgamma, float* dbeta, ccd_bf16* dx, long lddx, long rows, int C, void* stream);
/* Classifier Conv2d(C, 2, 3, padding=1) (segmentor.py:86) factored through pixel-wise GEMMs:
 *   forward : zT[co*9+tap, q] = sum_c w[co,c,tap] x[q,c] by ccd_gemm_nt (fp32, row stride ldz), then
 *             logits[n,co,y,x] = bias[co] + sum_tap zT[co*9+tap, (n,y+dy,x+dx)]            (ccd_cls_gather_fwd)
 *   backward: g[q, co*9+tap] = dlogits[n,co,(y,x)-(dy,dx)], bf16 [pixels, 64], columns >= 18 zero (ccd_cls_grad_cols);
 *             dx = g . Wd^T (ccd_gemm_nt), dW = g^T . x (ccd_gemm_tn), db = column sums of the centre taps. */
int ccd_cls_gather_fwd(const float* zT, long ldz, const float* bias, float* logits, int images, int H, int W,
                       void* stream);
int ccd_cls_grad_cols(const float* dlogits, ccd_bf16* g, int images, int H, int W, void* stream);
/* The same tail - unpool2's BatchNorm2d + ReLU, then `cls` (segmentor.py:84-95) - FUSED for the reference's own head shape
 * (C = 128 channels, H x W = 32 x 128 pixels per image, 2 classes; anything else: CCD_ESHAPE, take the kernels above).
 * y: the transposed conv's output before its BatchNorm, bf16 [images*H*W, ldy]; w: cls.weight fp32 [2, C, 3, 3] (the
 * parameter itself, no re-laid operand); relu(bn(y)) is never materialised (kernels/cls_tail.h).
 *   ccd_cls_tail_fwd        : logits fp32 [images, 2, H, W] = cls(relu(bn(y)))
 *   ccd_cls_tail_bwd_reduce : red[0:C] += sum d*[a>0], red[C:2C] += sum d*[a>0]*xhat with d = d(a) of cls computed on
 *                             the fly from dlogits fp32 [images, 2, H, W]; db_cls[2] += sum dlogits
 *   ccd_cls_tail_bwd_apply  : dy bf16 [images*H*W, lddy] = gamma*rstd*(d*[a>0] - red0/count - xhat*red1/count)  (red: summed
 *                             over the ranks by the caller); dgamma += red_local[C:2C], dbeta += red_local[0:C];
 *                             dbias_t[C] += column sums of dy (the transposed conv's bias gradient);
 *                             dw_cls fp32 [2, C, 3, 3] += sum_pixels g (x) a                                              */
int ccd_cls_tail_fwd(const ccd_bf16* y, long ldy, const float* mean_rstd, const float* gamma, const float* beta,
                     const float* w, const float* bias, float* logits, int images, int H, int W, int C, void* stream);
int ccd_cls_tail_bwd_reduce(const float* dlogits, const ccd_bf16* y, long ldy, const float* mean_rstd, const float* gamma,
                            const float* beta, const float* w, float* red, float* db_cls, int images, int H, int W, int C,
                            void* stream);
int ccd_cls_tail_bwd_apply(const float* dlogits, const ccd_bf16* y, long ldy, const float* mean_rstd, const float* gamma,
                           const float* beta, const float* w, const float* red, float count, const float* red_local,
                           float* dgamma, float* dbeta, float* dw_cls, float* dbias_t, ccd_bf16* dy, long lddy, int images,
                           int H, int W, int C, void* stream);
/* dst[sum_i idx_i*dst_strides[i]] <- src[sum_i idx_i*src_strides[i]] over dims[4] (host arrays); accumulate = 0: dst
 * is bf16 (cast), accumulate = 1: dst is fp32 and += (conv weights <-> GEMM operand layouts, weight gradients back). */
int ccd_permute4(const float* src, const long* src_strides, const long* dst_strides, const int* dims, void* dst,
                 int accumulate, void* stream);

/* Several re-layouts / BatchNorm finalisations in ONE launch (the segmentation head issued 27 + 16 launches of ~5 us per step for
 * them).  jobs: host array, n <= 24 (permute) / n <= 4 (finalize; the layers of one level, whose statistics arrive together).
 * ccd_bn_finalize_multi also adds 1 to *batches (nn.BatchNorm2d.num_batches_tracked, int64 on the device) where given. */
typedef struct { const float* src; void* dst; long src_strides[4]; long dst_strides[4]; int dims[4]; } ccd_permute4_job;
int ccd_permute4_multi(const ccd_permute4_job* jobs, int n, int accumulate, void* stream);
typedef struct {
    const float* stats; float* mean_rstd; float* running_mean; float* running_var; long* batches;
    float count, eps, momentum; int C;
} ccd_bn_finalize_job;
int ccd_bn_finalize_multi(const ccd_bn_finalize_job* jobs, int n, void* stream);

/* ---- finetune path: DINO_Finetune = ViT encoder + Mlp + NRTR decoder + TFLoss ---------------
 * Reference: Dino/model/dino_vision.py:134-246, Dino/decoder/nrtr_decoder.py:92-170, transformer_module.py:8-97,
 * transformer_layers.py:150-163, Dino/loss/ce_loss.py:94-128, train_finetune.py:262-289.  The Linear layers run on
 * ccd_gemm_nt / ccd_gemm_tn, the LayerNorms on ccd_ln_fwd / ccd_ln_bwd. */
/* nn.Dropout(p) without a mask tensor: element i is kept iff hash(seed, i) >= p * 2^32 and scaled by 1/(1-p); the
 * backward pass calls the same function on the gradient with the same seed.  dst = (resid ? resid : 0) + drop(src).
 * src / dst: fp32 or bf16 (flags), resid fp32 or NULL; n % 4 == 0; dst may alias src. */
int ccd_dropout(const void* src, int src_bf16, const float* resid, void* dst, int dst_bf16, long n, uint64_t seed, float p,
                void* stream);
/* DropPath scales of one backbone pass (vision_transformer.py:27-35,107-113): out[blk*per_block + j] = keep_j / keep[blk]
 * with keep_j ~ Bernoulli(keep[blk]) from the same counter-based generator (keep: device array [nblocks]). */
/* d_seed (optional, device): added to `seed` at run time (HIP-graph replays draw new masks) */
int ccd_droppath_scales(const float* keep, float* out, int per_block, int nblocks, uint64_t seed, const uint64_t* d_seed,
                        void* stream);
/* x[r,:] = dropout(trg_word_emb[tokens[r]] + position_table[r % T])   (nrtr_decoder.py:93-95); D % 4 == 0 */
int ccd_dec_embed_fwd(const int64_t* tokens, const float* emb, const float* pos, float* x, int rows, int T, int D,
                      int num_classes, uint64_t seed, float p, void* stream);
/* demb[c,:] += sum_{r: tokens[r]==c} drop(dx[r,:]) for c != padding_idx (nn.Embedding(padding_idx)); D <= 1024 */
int ccd_dec_embed_bwd(const int64_t* tokens, const float* dx, float* demb, int rows, int D, int num_classes, int padding_idx,
                      uint64_t seed, float p, void* stream);
/* MultiHeadAttention core (transformer_module.py:22-32, 84-92) for Tq <= 32 queries, Tk <= 256 keys, d_k = d_v = 64:
 *   out[b*Tq+t, 64h:64h+64] = dropout(softmax(mask(scale * q.k^T))) . v      lse[b,h,t] saved for the backward pass
 * q rows b*Tq+t (stride ldq), k / v rows b*Tk+j (strides ldk / ldv), head h at column 64h of each.  Mask: key j is
 * hidden from query t if causal && j > t, if tokens && tokens[b,j] == pad_idx (get_pad_mask & get_subsequent_mask,
 * nrtr_decoder.py:77-90), or if key_len && j >= key_len[b] (_get_mask :113-127).  probs (optional, fp32
 * [B,H,Tq,Tk]) receives the attention weights after dropout - what MultiHeadAttention returns as `attn`. */
int ccd_dec_attn_fwd(const ccd_bf16* q, long ldq, const ccd_bf16* k, long ldk, const ccd_bf16* v, long ldv, ccd_bf16* out,
                     long ldo, float* lse, float* probs, const int64_t* tokens, const int* key_len, int pad_idx, int causal,
                     int B, int H, int Tq, int Tk, float scale, uint64_t seed, float p, void* stream);
int ccd_dec_attn_bwd(const ccd_bf16* q, long ldq, const ccd_bf16* k, long ldk, const ccd_bf16* v, long ldv,
                     const ccd_bf16* out, const ccd_bf16* d_out, long ldo, const float* lse, const int64_t* tokens,
                     const int* key_len, int pad_idx, int causal, int B, int H, int Tq, int Tk, float scale, uint64_t seed,
                     float p, ccd_bf16* dq, long lddq, ccd_bf16* dk, long lddk, ccd_bf16* dv, long lddv, void* stream);
/* TFLoss (ce_loss.py:94-128): rows r = (b,t) of logits [B*T, ldl] (C <= 128 classes) against targets[b,t+1]; rows with
 * t == T-1 or target == pad_idx do not count.  fwd: acc[0] = sum of -log softmax[target], acc[1] = count (acc is
 * cleared first), row_lse[r] saved; the loss is acc[0]/acc[1].  bwd: d_logits (bf16 [B*T, ldd], zero beyond C) =
 * (softmax - onehot) * upstream[0] / count (upstream: device scalar, NULL = 1). */
int ccd_tf_loss_fwd(const float* logits, long ldl, int C, const int64_t* targets, int rows, int T, int pad_idx,
                    float* row_lse, float* acc, void* stream);
int ccd_tf_loss_bwd(const float* logits, long ldl, int C, const int64_t* targets, int rows, int T, int pad_idx,
                    const float* row_lse, const float* acc, const float* upstream, ccd_bf16* d_logits, long ldd, void* stream);
/* One greedy decoding position (nrtr_decoder.py:160-168): probs[b,step,:] = softmax(logits[b,:C]),
 * seq[b,step+1] = argmax (first maximum). */
int ccd_greedy_step(const float* logits, long ldl, int C, int B, float* probs, int steps, int step, int64_t* seq,
                    int seq_len, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCD_HIP_H */
