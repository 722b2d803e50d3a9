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
// one workgroup = WN_ROWS rows; a wave reads a row ONCE with 16-byte loads (round 5: two passes of 4-byte loads and 64-byte segments
// of the transposed copy ran at 1.3 TB/s), the transposed copy leaves in 4-byte pairs of rows = 128-byte segments
constexpr int WN_ROWS = 64, WN_MAX_V4 = 4;                 // D <= 64 lanes * 4 floats * WN_MAX_V4
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             bf16_t* __restrict__ w, bf16_t* __restrict__ w_t,
                                                             float* __restrict__ inv_out, int K, int D) {
    float* tile = reinterpret_cast<float*>(dynamic_smem());      // [WN_ROWS][D + 1]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k0 = blockIdx.x * WN_ROWS;
    const int d4 = D >> 2;
    for (int r = wv; r < WN_ROWS; r += 4) {
        const int k = k0 + r;
        if (k >= K) break;
        const f32x4v* row = reinterpret_cast<const f32x4v*>(v + (long)k * D);
        f32x4v x[WN_MAX_V4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < WN_MAX_V4; ++j) {
            const int c4 = lane + 64 * j;
            x[j] = c4 < d4 ? row[c4] : f32x4v{0.f, 0.f, 0.f, 0.f};
            s += x[j].x * x[j].x + x[j].y * x[j].y + x[j].z * x[j].z + x[j].w * x[j].w;
        }
        const float inv = 1.0f / sqrtf(wave_sum(s));
        const float sc = g[k] * inv;
#pragma unroll
        for (int j = 0; j < WN_MAX_V4; ++j) {
            const int c4 = lane + 64 * j;
            if (c4 < d4) {
                const float a0 = x[j].x * sc, a1 = x[j].y * sc, a2 = x[j].z * sc, a3 = x[j].w * sc;
                u32x2 o;
                o.x = pack_bf2(a0, a1); o.y = pack_bf2(a2, a3);
                *reinterpret_cast<u32x2*>(w + (long)k * D + 4 * c4) = o;
                float* t = tile + r * (D + 1) + 4 * c4;
                t[0] = a0; t[1] = a1; t[2] = a2; t[3] = a3;
            }
        }
        if (lane == 0) inv_out[k] = inv;
    }
    __syncthreads();
    if (w_t) {
        const int kp = threadIdx.x & 31;                         // rows 2 kp, 2 kp + 1 of the tile -> one 4-byte store
        for (int c = threadIdx.x >> 5; c < D; c += 8)
            if (k0 + 2 * kp + 1 < K)
                *reinterpret_cast<unsigned*>(w_t + (long)c * K + k0 + 2 * kp) =
                    pack_bf2(tile[(2 * kp) * (D + 1) + c], tile[(2 * kp + 1) * (D + 1) + c]);
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
