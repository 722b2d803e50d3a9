// layernorm.h - LayerNorm(eps) over the fp32 residual stream, bf16 out (vision_transformer.py:99,103,156,162).
// One wave per token row; a lane keeps its <= 8 elements in registers (E <= 512), statistics by wave shuffles.
// HBM-bound: algorithmic bytes per row = 4*E (x) + 2*E (y) forward; 4*E (x) + 2*E (dy) + 8*E (g r/w) backward.
#pragma once

namespace ccd {

constexpr int LN_MAX_PER_LANE = 8;

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int rows, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                       // whole waves exit together; no block barrier below
    const float* xr = x + (long)row * E;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < E ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = c < E ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
    bf16_t* yr = y + (long)row * E;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < E) yr[c] = f2bf((v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

// g[row,:] (+)= LN_backward(dy[row,:]) ; dgamma += sum_rows dy*xhat ; dbeta += sum_rows dy   (fp32 atomics)
// ACCUM = true : g += dx   (residual-gradient stream of a transformer block)
// ACCUM = false: g  = dx   (taps / final norm feeding a fresh stream)
template <bool ACCUM>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, float* __restrict__ g,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                     int E, int rows_per_block) {
    __shared__ float red[2][4][64 * LN_MAX_PER_LANE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row_begin = blockIdx.x * rows_per_block;
    const int row_end = row_begin + rows_per_block < rows ? row_begin + rows_per_block : rows;
    float gam[LN_MAX_PER_LANE], dg[LN_MAX_PER_LANE], db[LN_MAX_PER_LANE];
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        gam[i] = c < E ? gamma[c] : 0.f;
        dg[i] = 0.f;
        db[i] = 0.f;
    }
    for (int row = row_begin + w; row < row_end; row += 4) {
        const float mu = mean[row], rs = rstd[row];
        const float* xr = x + (long)row * E;
        const bf16_t* dyr = dy + (long)row * E;
        float xh[LN_MAX_PER_LANE], d[LN_MAX_PER_LANE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < E;
            xh[i] = ok ? (xr[c] - mu) * rs : 0.f;
            const float dyv = ok ? bf2f(dyr[c]) : 0.f;
            d[i] = dyv * gam[i];
            s1 += d[i];
            s2 += d[i] * xh[i];
            dg[i] += dyv * xh[i];
            db[i] += dyv;
        }
        s1 = wave_sum(s1) / (float)E;
        s2 = wave_sum(s2) / (float)E;
        float* gr = g + (long)row * E;
#pragma unroll
        for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < E) {
                const float dx = rs * (d[i] - s1 - xh[i] * s2);
                gr[c] = ACCUM ? gr[c] + dx : dx;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        red[0][w][lane + 64 * i] = dg[i];
        red[1][w][lane + 64 * i] = db[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) {
        const float a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        atomicAdd(dgamma + c, a);
        atomicAdd(dbeta + c, b);
    }
}

}  // namespace ccd
