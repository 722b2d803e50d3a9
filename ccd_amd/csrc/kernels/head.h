// head.h - the non-GEMM pieces of DINOHead (Dino/modules/vision_transformer.py:294-328):
//   F.normalize(x, dim=-1, p=2)  (eps 1e-12)                                      :326
//   weight_norm(Linear(bottleneck, out_dim, bias=False)):  w = g * v / ||v||_row  :313
// The Linear layers themselves run on the MFMA GEMM (gemm.h).  Row counts that depend on the number of
// selected character rows are read from device memory (d_rows) so the host never synchronises.
#pragma once

namespace ccd {

constexpr int HD_MAX_PER_LANE = 8;     // bottleneck dim <= 512

// y = x / max(||x||, 1e-12) per row (bf16 in/out), inv[row] saved
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                         float* __restrict__ inv_out, int max_rows,
                                                         const int* __restrict__ d_rows, int rows_mul, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = d_rows ? d_rows[0] * rows_mul : max_rows;
    if (row >= rows || row >= max_rows) return;
    float v[HD_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HD_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < D ? bf2f(x[(long)row * D + c]) : 0.f;
        s += v[i] * v[i];
    }
    const float nrm = sqrtf(wave_sum(s));
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
#pragma unroll
    for (int i = 0; i < HD_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < D) y[(long)row * D + c] = f2bf(v[i] * inv);
    }
    if (lane == 0) inv_out[row] = inv;
}
// dx = inv * (dy - y * <y, dy>),  y = x * inv
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ inv_in,
                                                         const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                                         int max_rows, const int* __restrict__ d_rows, int rows_mul,
                                                         int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rows = d_rows ? d_rows[0] * rows_mul : max_rows;
    if (row >= rows || row >= max_rows) return;
    const float inv = inv_in[row];
    float yv[HD_MAX_PER_LANE], dv[HD_MAX_PER_LANE];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < HD_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        yv[i] = c < D ? bf2f(x[(long)row * D + c]) * inv : 0.f;
        dv[i] = c < D ? bf2f(dy[(long)row * D + c]) : 0.f;
        dot += yv[i] * dv[i];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < HD_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < D) dx[(long)row * D + c] = f2bf(inv * (dv[i] - yv[i] * dot));
    }
}

// effective last-layer weight: w[k,:] = g[k] * v[k,:] / ||v[k,:]||  -> bf16 [K,D] and bf16 transposed [D,K]
// one workgroup = 32 rows
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             bf16_t* __restrict__ w, bf16_t* __restrict__ w_t,
                                                             float* __restrict__ inv_out, int K, int D) {
    float* tile = reinterpret_cast<float*>(dynamic_smem());      // [32][D + 1]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k0 = blockIdx.x * 32;
    for (int r = wv; r < 32; r += 4) {
        const int k = k0 + r;
        if (k >= K) break;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) { const float a = v[(long)k * D + c]; s += a * a; }
        const float inv = 1.0f / sqrtf(wave_sum(s));
        const float sc = g[k] * inv;
        for (int c = lane; c < D; c += 64) {
            const float a = v[(long)k * D + c] * sc;
            w[(long)k * D + c] = f2bf(a);
            tile[r * (D + 1) + c] = a;
        }
        if (lane == 0) inv_out[k] = inv;
    }
    __syncthreads();
    if (w_t) {
        const int kk = threadIdx.x & 31;
        for (int c = threadIdx.x >> 5; c < D; c += 8)
            if (k0 + kk < K) w_t[(long)c * K + k0 + kk] = f2bf(tile[kk * (D + 1) + c]);
    }
}
// dv = g*inv * (dw - vhat <dw, vhat>), dg = <dw, vhat>, vhat = v*inv.  Grad slots are overwritten.
__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             const float* __restrict__ inv_in,
                                                             const float* __restrict__ dw, float* __restrict__ dv,
                                                             float* __restrict__ dg, int K, int D) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    const float inv = inv_in[k], gk = g[k];
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot += dw[(long)k * D + c] * v[(long)k * D + c] * inv;
    dot = wave_sum(dot);
    for (int c = lane; c < D; c += 64) {
        const float vh = v[(long)k * D + c] * inv;
        dv[(long)k * D + c] = gk * inv * (dw[(long)k * D + c] - vh * dot);
    }
    if (dg && lane == 0) dg[k] = dot;
}

}  // namespace ccd
