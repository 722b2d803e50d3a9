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

/* ---------------------------------------------------------------- GEMM family (nn.Linear / conv-as-GEMM)
 * epilogue codes (ccd_gemm_*: `epilogue`):                                                              */
#define CCD_EPI_BF16 0     /* C(bf16) = acc + bias                                                       */
#define CCD_EPI_GELU 1     /* C(bf16) = u = acc + bias, C2(bf16) = gelu(u)   Mlp.fc1+act, vit.py:59-61    */
#define CCD_EPI_RESID 2    /* C(f32) = resid + (acc + bias) * rowscale[row / rows_per_sample]  Block :109 */
#define CCD_EPI_F32 3      /* C(f32) = acc + bias                                                        */
#define CCD_EPI_ATOMIC 4   /* C(f32) += acc   (split-K partial sums)                                     */
#define CCD_EPI_DGELU 5    /* C(bf16) = acc * gelu'(aux)                                                 */

/* C[M,N] = A[M,K] . B[N,K]^T   (F.linear(x, W): Dino/modules/vision_transformer.py:60,63,82,90,325-327)
 * K % 64 == 0, N % 8 == 0.  m_fastest: tile order hint (1 when B is the large operand). */
int ccd_gemm_nt(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int M, int N, int K, int epilogue,
                void* C, long ldc, void* C2, long ldc2, const float* bias, const float* resid, long ldr,
                const float* rowscale, int rows_per_sample, const ccd_bf16* aux, long ldaux, float alpha,
                int m_fastest, void* stream);
/* C[P,Q] (+)= sum_m A[m,P] * B[m,Q]   (weight gradients dW = dY^T X of every Linear; autograd of the above)
 * P % 8 == 0, Q % 8 == 0; epilogue CCD_EPI_ATOMIC accumulates into fp32 C (split over m), CCD_EPI_F32 stores. */
int ccd_gemm_tn(const ccd_bf16* A, long lda, const ccd_bf16* B, long ldb, int P, int Q, int Mc, int epilogue,
                float* C, long ldc, float alpha, int splits, void* stream);

/* ---------------------------------------------------------------- LayerNorm (eps 1e-6), vit.py:99,103,156,162-166 */
int ccd_ln_fwd(const float* x, const float* gamma, const float* beta, ccd_bf16* y, float* mean, float* rstd,
               int rows, int E, float eps, void* stream);
/* g (+)= dx ; dgamma += ..., dbeta += ... (fp32 atomics; caller zeroes them once per step) */
int ccd_ln_bwd(const ccd_bf16* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
               float* g, int accumulate, float* dgamma, float* dbeta, int rows, int E, void* stream);

/* ---------------------------------------------------------------- attention, vit.py:80-92 (T = 256, head_dim = 64)
 * qkv [views, 256, 3, heads, 64] bf16 (the layout Attention.forward reshapes to), out [views, 256, heads*64]  */
int ccd_attention_fwd(const ccd_bf16* qkv, ccd_bf16* out, float* lse, int views, int heads, float scale,
                      void* stream);
int ccd_attention_bwd(const ccd_bf16* qkv, const ccd_bf16* out, const ccd_bf16* d_out, const float* lse,
                      float* delta_ws, ccd_bf16* d_qkv, int views, int heads, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCD_HIP_H */
