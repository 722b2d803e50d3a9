// conv.h - the non-GEMM pieces of the segmentation head (Dino/modules/segmentor.py:38-95) on channels-last bf16
// activations [pixels, C]:
//   im2col_kernel              explicit patch matrix (only for the WEIGHT gradients; forward / data gradients of the
//                              3x3 convs and transposed convs run as implicit GEMMs through gemm.h's gather loader)
//   bn_finalize / bn_relu_fwd / bn_relu_bwd_reduce / bn_relu_bwd_apply
//                              train-mode BatchNorm2d (+ReLU) with batch statistics taken from the GEMM epilogue
//                              (column sum / sum of squares); the cross-rank reduction of SyncBatchNorm happens on the
//                              host between *_reduce/finalize and *_apply (two small all-reduces per layer)
//   cls_gather_fwd / cls_grad_cols
//                              the final 3x3 conv 128 -> 2 classes factored through pixel-wise GEMMs (see below),
//                              fp32 logits in the reference's NCHW layout
#pragma once

namespace ccd {

typedef ::ccd_conv_desc ConvDesc;  // include/ccd_hip.h; mirrors the gather fields of GemmParams

// cols[r, tap*cin + c] = src[(n, oy*s_mul + dy[tap], ox*s_mul + dx[tap]), c] or 0; one 16-byte chunk per thread
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ src, long src_stride, ConvDesc d,
                                                     long rows, bf16_t* __restrict__ cols) {
    const int chunks_per_tap = d.cin >> 3;
    const long per_row = (long)d.ntaps * chunks_per_tap;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row) return;
    const long r = i / per_row;
    const int q = (int)(i % per_row), tap = q / chunks_per_tap, c = (q % chunks_per_tap) * 8;
    const int hw = d.g_h_log2 + d.g_w_log2;
    const int n = (int)(r >> hw), oy = (int)(r >> d.g_w_log2) & ((1 << d.g_h_log2) - 1), ox = (int)r & ((1 << d.g_w_log2) - 1);
    const int sy = oy * d.s_mul + d.dy[tap], sx = ox * d.s_mul + d.dx[tap];
    u32x4 v = {0u, 0u, 0u, 0u};
    if (sy >= 0 && sy < d.s_h && sx >= 0 && sx < d.s_w)
        v = *reinterpret_cast<const u32x4*>(src + ((long)n * d.s_h * d.s_w + (long)sy * d.s_w + sx) * src_stride + c);
    *reinterpret_cast<u32x4*>(cols + r * per_row * 8 + (long)tap * d.cin + c) = v;
}

// ------------------------------------------------------------------------------------------------- BatchNorm
// stats[0:C] = sum x, stats[C:2C] = sum x^2 over `count` pixels (already reduced over ranks by the host)
__global__ void bn_finalize_kernel(const float* __restrict__ stats, float count, float eps, float momentum,
                                   float* __restrict__ mean_rstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[c] / count;
    float var = stats[C + c] / count - mean * mean;
    var = var > 0.f ? var : 0.f;
    mean_rstd[c] = mean;
    mean_rstd[C + c] = 1.0f / sqrtf(var + eps);
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * var * (count / (count - 1.0f));
}

// The same for up to BN_MULTI_MAX layers in ONE launch (blockIdx.y = layer), which also counts the batch (num_batches_tracked += 1):
// the head has 8 BatchNorm layers in 4 dependent groups, and a 5-us launch per layer and per counter was 16 launches of a 1.6-ms forward.
constexpr int BN_MULTI_MAX = 4;
struct BnFinalizeJob {
    const float* stats; float* mean_rstd; float* running_mean; float* running_var; long* batches;
    float count, eps, momentum; int C;
};
struct BnFinalizeJobs { BnFinalizeJob j[BN_MULTI_MAX]; };
__global__ void bn_finalize_multi_kernel(BnFinalizeJobs jobs) {
    const BnFinalizeJob& q = jobs.j[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && q.batches) *q.batches += 1;
    if (c >= q.C) return;
    const float mean = q.stats[c] / q.count;
    float var = q.stats[q.C + c] / q.count - mean * mean;
    var = var > 0.f ? var : 0.f;
    q.mean_rstd[c] = mean;
    q.mean_rstd[q.C + c] = 1.0f / sqrtf(var + q.eps);
    q.running_mean[c] = (1.0f - q.momentum) * q.running_mean[c] + q.momentum * mean;
    q.running_var[c] = (1.0f - q.momentum) * q.running_var[c] + q.momentum * var * (q.count / (q.count - 1.0f));
}

// y = relu((x - mean) * rstd * gamma + beta); x [rows, ldx], y [rows, ldy] bf16; 8 channels per thread.
// A thread walks chunks 256 apart (a multiple of C / 8: it keeps ITS 8 channels, their 32 parameters are loaded once; the row
// index advances by 256 / (C / 8) - as one chunk per thread the kernel did a 64-bit division and 32 parameter loads per 16-byte
// data load and ran at 2.0 TB/s), four chunks in flight per thread.
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const bf16_t* __restrict__ x, long ldx,
                                                          const float* __restrict__ mean_rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16_t* __restrict__ y, long ldy, long rows, int C) {
    const int c8 = C >> 3;
    // a block streams ONE contiguous range of chunks (a multiple of 256, so a thread stays on its 8 channels): linear walks
    // reach the HBM rate, a grid-wide stride (16 MB between a thread's chunks) stayed at 2.6 TB/s
    const long total = rows * c8, T = 256;
    const long per_block = (((total + gridDim.x - 1) / gridDim.x + 255) / 256) * 256;
    long i = (long)blockIdx.x * per_block + threadIdx.x;
    const long end = (i - threadIdx.x + per_block) < total ? (i - threadIdx.x + per_block) : total;
    if (i >= end) return;
    const int c = (int)(i % c8) * 8;
    float mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = mean_rstd[c + e]; rs[e] = mean_rstd[C + c + e]; ga[e] = gamma[c + e]; be[e] = beta[c + e]; }
    auto one = [&](const u32x4& w) {
        float v[8];
        unpack8(w, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float o = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];
            v[e] = o > 0.f ? o : 0.f;
        }
        return pack8(v);
    };
    // (the row index advances by T / c8 per stride: no 64-bit division per chunk - that alone held the kernel at 2 TB/s)
    long r = i / c8;
    const long dr = T / c8;
    for (; i + 3 * T < end; i += 4 * T, r += 4 * dr) {
        u32x4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const u32x4*>(x + (r + u * dr) * ldx + c);
        // (all four have arrived before the first store: left alone the scheduler sinks every load next to its use - one load
        // in flight per thread, each behind `vmcnt(0)`, i.e. behind the previous store's round trip: 2.6 TB/s instead of 6)
#pragma unroll
        for (int u = 0; u < 4; ++u) needed_here(w[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(y + (r + u * dr) * ldy + c) = one(w[u]);
    }
    for (; i < end; i += T, r += dr) *reinterpret_cast<u32x4*>(y + r * ldy + c) = one(*reinterpret_cast<const u32x4*>(x + r * ldx + c));
}

// red[0:C] += sum dy*[y>0], red[C:2C] += sum dy*[y>0]*xhat     (dy [rows, lddy] bf16); geometry as colsum_bf16
__global__ __launch_bounds__(COLSUM_THREADS) void bn_relu_bwd_reduce_kernel(const bf16_t* __restrict__ dy, long lddy,
                                                                 const bf16_t* __restrict__ x, long ldx,
                                                                 const float* __restrict__ mean_rstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ red,
                                                                 long rows, int C, int rows_per_block, int cgn_log2) {
    __shared__ float part[2][COLSUM_THREADS][8];
    const int cgn = 1 << cgn_log2, rln = COLSUM_THREADS >> cgn_log2;
    const int cg = threadIdx.x & (cgn - 1), rl = threadIdx.x >> cgn_log2;
    const int c = (blockIdx.x * cgn + cg) * 8;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float mu[8], rs[8], ga[8], be[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean_rstd[c + e]; rs[e] = mean_rstd[C + c + e]; ga[e] = gamma[c + e]; be[e] = beta[c + e]; }
        auto accumulate = [&](const u32x4& xw, const u32x4& dw) {
            float xv[8], dv[8];
            unpack8(xw, xv);
            unpack8(dw, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                const float d = (xh * ga[e] + be[e]) > 0.f ? dv[e] : 0.f;
                s1[e] += d;
                s2[e] += d * xh;
            }
        };
        long r = r0 + rl;
        for (; r + 3 * rln < r1; r += 4 * rln) {
            u32x4 xw[4], dw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xw[u] = *reinterpret_cast<const u32x4*>(x + (r + u * rln) * ldx + c);
                dw[u] = *reinterpret_cast<const u32x4*>(dy + (r + u * rln) * lddy + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accumulate(xw[u], dw[u]);
        }
        for (; r < r1; r += rln)
            accumulate(*reinterpret_cast<const u32x4*>(x + r * ldx + c), *reinterpret_cast<const u32x4*>(dy + r * lddy + c));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[0][threadIdx.x][e] = s1[e]; part[1][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    if ((int)threadIdx.x < 8 * cgn) {                        // coalesced publication, one column per thread
        const int c_local = threadIdx.x, g = c_local >> 3, e = c_local & 7;
        const int cc = blockIdx.x * cgn * 8 + c_local;
        if (cc < C) {
            float a = 0.f, b = 0.f;
            for (int j = 0; j < rln; ++j) { a += part[0][j * cgn + g][e]; b += part[1][j * cgn + g][e]; }
            atomicAdd(red + cc, a);
            atomicAdd(red + C + cc, b);
        }
    }
}

// dx = gamma*rstd*(dy*[y>0] - red0/count - xhat*red1/count)      (red already summed over ranks); grid-stride as bn_relu_fwd
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const bf16_t* __restrict__ dy, long lddy,
                                                                const bf16_t* __restrict__ x, long ldx,
                                                                const float* __restrict__ mean_rstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ red, float count,
                                                                const float* __restrict__ red_local,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                bf16_t* __restrict__ dx, long lddx, long rows, int C) {
    if (blockIdx.x == 0 && (int)threadIdx.x < C) {     // parameter gradients from this rank's own sums (C <= 256)
        dbeta[threadIdx.x] += red_local[threadIdx.x];
        dgamma[threadIdx.x] += red_local[C + threadIdx.x];
    }
    const int c8 = C >> 3;
    const long total = rows * c8, T = 256;
    const long per_block = (((total + gridDim.x - 1) / gridDim.x + 255) / 256) * 256;
    long i = (long)blockIdx.x * per_block + threadIdx.x;
    const long end = (i - threadIdx.x + per_block) < total ? (i - threadIdx.x + per_block) : total;
    if (i >= end) return;
    const int c = (int)(i % c8) * 8;
    const float inv = 1.0f / count;
    float mu[8], rs[8], ga[8], be[8], k0[8], k1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mu[e] = mean_rstd[c + e]; rs[e] = mean_rstd[C + c + e]; ga[e] = gamma[c + e]; be[e] = beta[c + e];
        k0[e] = red[c + e] * inv; k1[e] = red[C + c + e] * inv;
    }
    auto one = [&](const u32x4& xw, const u32x4& dw) {
        float xv[8], dv[8], o[8];
        unpack8(xw, xv);
        unpack8(dw, dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            const float d = (xh * ga[e] + be[e]) > 0.f ? dv[e] : 0.f;
            o[e] = ga[e] * rs[e] * (d - k0[e] - xh * k1[e]);
        }
        return pack8(o);
    };
    long r = i / c8;
    const long dr = T / c8;
    for (; i + 1 * T < end; i += 2 * T, r += 2 * dr) {      // (two chunks of x and dy = four 16-byte loads in flight)
        u32x4 xw[2], dw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            xw[u] = *reinterpret_cast<const u32x4*>(x + (r + u * dr) * ldx + c);
            dw[u] = *reinterpret_cast<const u32x4*>(dy + (r + u * dr) * lddy + c);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) { needed_here(xw[u]); needed_here(dw[u]); }
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<u32x4*>(dx + (r + u * dr) * lddx + c) = one(xw[u], dw[u]);
    }
    for (; i < end; i += T, r += dr)
        *reinterpret_cast<u32x4*>(dx + r * lddx + c) = one(*reinterpret_cast<const u32x4*>(x + r * ldx + c),
                                                           *reinterpret_cast<const u32x4*>(dy + r * lddy + c));
}

// ------------------------------------------------------------------------------- classifier conv 3x3, C -> 2
// N = 2 output channels is no MFMA tile, but the conv factors through a pixel-wise GEMM:
//   forward : zT[co*9+tap, q] = sum_c w[co,c,tap] x[q,c]  (plain NT GEMM, 18 -> 32 rows, fp32 out), then
//             logits[n,co,y,x] = bias[co] + sum_tap zT[co*9+tap, (n, y+dy, x+dx)]          (cls_gather_fwd_kernel)
//   backward: g[q, co*9+tap] = dlogits[n, co, (y,x) - (dy,dx)]                              (cls_grad_cols_kernel)
//             dx = g . Wd^T (NT GEMM, K = 64), dW = g^T . x (TN GEMM), db = column sums of g's centre taps.
constexpr int CLS_ZROWS = 32, CLS_GCOLS = 64;

__global__ __launch_bounds__(256) void cls_gather_fwd_kernel(const float* __restrict__ zT, long ldz,
                                                             const float* __restrict__ bias, float* __restrict__ logits,
                                                             int images, int H, int W) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)images * H * W) return;
    const int n = (int)(p / (H * W)), yy = (int)(p / W) % H, xx = (int)(p % W);
    float a0 = bias[0], a1 = bias[1];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int sy = yy + dy, sx = xx + dx;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            const long q = p + dy * W + dx;
            a0 += zT[tap * ldz + q];
            a1 += zT[(9 + tap) * ldz + q];
        }
    }
    const long plane = (long)H * W;
    logits[((long)n * 2) * plane + (long)yy * W + xx] = a0;
    logits[((long)n * 2 + 1) * plane + (long)yy * W + xx] = a1;
}

// g [pixels, 64] bf16: column co*9+tap (< 18) = dlogits[n, co, y-dy, x-dx] (0 outside), columns 18..63 = 0
__global__ __launch_bounds__(256) void cls_grad_cols_kernel(const float* __restrict__ dl, bf16_t* __restrict__ g,
                                                            int images, int H, int W) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)images * H * W * 8) return;
    const long q = i >> 3;
    const int chunk = (int)(i & 7);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (chunk < 3) {
        const int n = (int)(q / (H * W)), yy = (int)(q / W) % H, xx = (int)(q % W);
        const long plane = (long)H * W;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = chunk * 8 + e;
            if (j < 18) {
                const int co = j / 9, tap = j % 9;
                const int sy = yy - (tap / 3 - 1), sx = xx - (tap % 3 - 1);
                if (sy >= 0 && sy < H && sx >= 0 && sx < W) v[e] = dl[((long)n * 2 + co) * plane + (long)sy * W + sx];
            }
        }
    }
    *reinterpret_cast<u32x4*>(g + q * CLS_GCOLS + chunk * 8) = pack8(v);
}

// ------------------------------------------------------------------------------------- weight re-layouts
// dst[i0*d0 + i1*d1 + i2*d2 + i3*d3] <- src[i0*s0 + i1*s1 + i2*s2 + i3*s3];  ACC = false: dst bf16 = cast(src),
// ACC = true: dst fp32 += src   (GEMM-operand views of conv weights; weight gradients back into parameter layout)
struct Permute4 {
    long s[4], d[4];
    int n[4];
};
template <bool ACC>
__global__ __launch_bounds__(256) void permute4_kernel(const float* __restrict__ src, Permute4 q, long total,
                                                       void* __restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int i3 = (int)(i % q.n[3]), i2 = (int)((i / q.n[3]) % q.n[2]), i1 = (int)((i / ((long)q.n[3] * q.n[2])) % q.n[1]);
    const long i0 = i / ((long)q.n[3] * q.n[2] * q.n[1]);
    const float v = src[i0 * q.s[0] + i1 * q.s[1] + i2 * q.s[2] + i3 * q.s[3]];
    const long o = i0 * q.d[0] + i1 * q.d[1] + i2 * q.d[2] + i3 * q.d[3];
    if (ACC) reinterpret_cast<float*>(dst)[o] += v;
    else reinterpret_cast<bf16_t*>(dst)[o] = f2bf(v);
}

// Up to PERMUTE_MULTI_MAX re-layouts in one launch (the head re-lays 22 weight tensors into GEMM operands per step and folds 5 staged
// weight gradients back: 27 launches of ~5 us); a block finds its job in the table of first blocks (wave-uniform scan).
constexpr int PERMUTE_MULTI_MAX = 24;
struct PermuteJob { const float* src; void* dst; Permute4 q; long total; unsigned first_block; };
struct PermuteJobs { PermuteJob j[PERMUTE_MULTI_MAX]; int n; };
template <bool ACC>
__global__ __launch_bounds__(256) void permute4_multi_kernel(PermuteJobs jobs) {
    int k = 0;
    while (k + 1 < jobs.n && blockIdx.x >= jobs.j[k + 1].first_block) ++k;
    const PermuteJob& jb = jobs.j[k];
    const Permute4& q = jb.q;
    const long i = (long)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (i >= jb.total) return;
    const int i3 = (int)(i % q.n[3]), i2 = (int)((i / q.n[3]) % q.n[2]), i1 = (int)((i / ((long)q.n[3] * q.n[2])) % q.n[1]);
    const long i0 = i / ((long)q.n[3] * q.n[2] * q.n[1]);
    const float v = jb.src[i0 * q.s[0] + i1 * q.s[1] + i2 * q.s[2] + i3 * q.s[3]];
    const long o = i0 * q.d[0] + i1 * q.d[1] + i2 * q.d[2] + i3 * q.d[3];
    if (ACC) reinterpret_cast<float*>(jb.dst)[o] += v;
    else reinterpret_cast<bf16_t*>(jb.dst)[o] = f2bf(v);
}

}  // namespace ccd
