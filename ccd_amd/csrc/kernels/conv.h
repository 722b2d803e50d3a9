// conv.h - the non-GEMM pieces of the segmentation head (Dino/modules/segmentor.py:38-95) on channels-last bf16
// activations [pixels, C]:
//   im2col_kernel              explicit patch matrix (only for the WEIGHT gradients; forward / data gradients of the
//                              3x3 convs and transposed convs run as implicit GEMMs through gemm.h's gather loader)
//   bn_finalize / bn_relu_fwd / bn_relu_bwd_reduce / bn_relu_bwd_apply
//                              train-mode BatchNorm2d (+ReLU) with batch statistics taken from the GEMM epilogue
//                              (column sum / sum of squares); the cross-rank reduction of SyncBatchNorm happens on the
//                              host between *_reduce/finalize and *_apply (two small all-reduces per layer)
//   cls_conv_fwd / cls_conv_bwd_data / cls_conv_bwd_weight
//                              the final 3x3 conv 128 -> 2 classes (N = 2 is no MFMA shape): VALU kernels,
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

// y = relu((x - mean) * rstd * gamma + beta); x [rows, ldx], y [rows, ldy] bf16; 8 channels per thread
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const bf16_t* __restrict__ x, long ldx,
                                                          const float* __restrict__ mean_rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16_t* __restrict__ y, long ldy, long rows, int C) {
    const int c8 = C >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8) return;
    const long r = i / c8;
    const int c = (int)(i % c8) * 8;
    float v[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + r * ldx + c), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float o = (v[e] - mean_rstd[c + e]) * mean_rstd[C + c + e] * gamma[c + e] + beta[c + e];
        v[e] = o > 0.f ? o : 0.f;
    }
    *reinterpret_cast<u32x4*>(y + r * ldy + c) = pack8(v);
}

// red[0:C] += sum dy*[y>0], red[C:2C] += sum dy*[y>0]*xhat     (dy [rows, lddy] bf16)
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const bf16_t* __restrict__ dy, long lddy,
                                                                 const bf16_t* __restrict__ x, long ldx,
                                                                 const float* __restrict__ mean_rstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ red,
                                                                 long rows, int C, int rows_per_block) {
    __shared__ float part[2][8][32][8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cg) * 8;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float mu[8], rs[8], ga[8], be[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean_rstd[c + e]; rs[e] = mean_rstd[C + c + e]; ga[e] = gamma[c + e]; be[e] = beta[c + e]; }
        for (long r = r0 + rl; r < r1; r += 8) {
            float xv[8], dv[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + r * ldx + c), xv);
            unpack8(*reinterpret_cast<const u32x4*>(dy + r * lddy + c), dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                const float d = (xh * ga[e] + be[e]) > 0.f ? dv[e] : 0.f;
                s1[e] += d;
                s2[e] += d * xh;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[0][rl][cg][e] = s1[e]; part[1][rl][cg][e] = s2[e]; }
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { a += part[0][j][cg][e]; b += part[1][j][cg][e]; }
            atomicAdd(red + c + e, a);
            atomicAdd(red + C + c + e, b);
        }
    }
}

// dx = gamma*rstd*(dy*[y>0] - red0/count - xhat*red1/count)      (red already summed over ranks)
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
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8) return;
    const long r = i / c8;
    const int c = (int)(i % c8) * 8;
    float xv[8], dv[8], o[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + r * ldx + c), xv);
    unpack8(*reinterpret_cast<const u32x4*>(dy + r * lddy + c), dv);
    const float inv = 1.0f / count;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float rs = mean_rstd[C + c + e], ga = gamma[c + e];
        const float xh = (xv[e] - mean_rstd[c + e]) * rs;
        const float d = (xh * ga + beta[c + e]) > 0.f ? dv[e] : 0.f;
        o[e] = ga * rs * (d - red[c + e] * inv - xh * red[C + c + e] * inv);
    }
    *reinterpret_cast<u32x4*>(dx + r * lddx + c) = pack8(o);
}

// ------------------------------------------------------------------------------- classifier conv 3x3, C -> 2
constexpr int CLS_MAX_C = 128;

// logits[n, co, y, x] (fp32 NCHW) = bias[co] + sum_{tap,c} x[(n, y+dy, x+dx), c] * w[co, c, tap];  w fp32 [2, C, 3, 3]
__global__ __launch_bounds__(256) void cls_conv_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ logits,
                                                           int images, int H, int W, int C) {
    __shared__ float ws[2 * 9 * CLS_MAX_C];                    // [co][tap][c]
    for (int i = threadIdx.x; i < 2 * 9 * C; i += 256) {
        const int co = i / (9 * C), tap = (i / C) % 9, c = i % C;
        ws[i] = w[((long)co * C + c) * 9 + tap];
    }
    __syncthreads();
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)images * H * W) return;
    const int n = (int)(p / (H * W)), yy = (int)(p / W) % H, xx = (int)(p % W);
    float a0 = bias[0], a1 = bias[1];
    for (int tap = 0; tap < 9; ++tap) {
        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
        if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
        const bf16_t* src = x + (((long)n * H + sy) * W + sx) * C;
        const float* w0 = ws + tap * C;
        const float* w1 = ws + (9 + tap) * C;
        for (int c = 0; c < C; c += 8) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(src + c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0 += v[e] * w0[c + e]; a1 += v[e] * w1[c + e]; }
        }
    }
    const long plane = (long)H * W;
    logits[((long)n * 2) * plane + (long)yy * W + xx] = a0;
    logits[((long)n * 2 + 1) * plane + (long)yy * W + xx] = a1;
}

// dx[(n,y,x), c] = sum_{tap,co} dl[n, co, y-dy, x-dx] * w[co, c, tap]        (8 channels per thread)
__global__ __launch_bounds__(256) void cls_conv_bwd_data_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                                bf16_t* __restrict__ dx, int images, int H, int W, int C) {
    __shared__ float ws[2 * 9 * CLS_MAX_C];
    for (int i = threadIdx.x; i < 2 * 9 * C; i += 256) {
        const int co = i / (9 * C), tap = (i / C) % 9, c = i % C;
        ws[i] = w[((long)co * C + c) * 9 + tap];
    }
    __syncthreads();
    const int c8 = C >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)images * H * W * c8) return;
    const long p = i / c8;
    const int c = (int)(i % c8) * 8;
    const int n = (int)(p / (H * W)), yy = (int)(p / W) % H, xx = (int)(p % W);
    const long plane = (long)H * W;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
        const int sy = yy - (tap / 3 - 1), sx = xx - (tap % 3 - 1);     // output pixel that read (y, x) through `tap`
        if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
        const float d0 = dl[((long)n * 2) * plane + (long)sy * W + sx], d1 = dl[((long)n * 2 + 1) * plane + (long)sy * W + sx];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += d0 * ws[tap * C + c + e] + d1 * ws[(9 + tap) * C + c + e];
    }
    *reinterpret_cast<u32x4*>(dx + p * C + c) = pack8(o);
}

// dw[co, c, tap] += sum_pixels dl[n, co, y, x] * x[(n, y+dy, x+dx), c]; db[co] += sum dl      (thread = (co, c))
__global__ __launch_bounds__(256) void cls_conv_bwd_weight_kernel(const float* __restrict__ dl, const bf16_t* __restrict__ x,
                                                                  float* __restrict__ dw, float* __restrict__ db,
                                                                  int images, int H, int W, int C, int pix_per_block) {
    const int co = threadIdx.x / C, c = threadIdx.x % C;      // blockDim = 2*C
    const long total = (long)images * H * W;
    const long p0 = (long)blockIdx.x * pix_per_block;
    const long p1 = p0 + pix_per_block < total ? p0 + pix_per_block : total;
    const long plane = (long)H * W;
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (long p = p0; p < p1; ++p) {
        const int n = (int)(p / (H * W)), yy = (int)(p / W) % H, xx = (int)(p % W);
        const float d = dl[((long)n * 2 + co) * plane + (long)yy * W + xx];
        bsum += d;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
            if (sy >= 0 && sy < H && sx >= 0 && sx < W) acc[tap] += d * bf2f(x[(((long)n * H + sy) * W + sx) * C + c]);
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) atomicAdd(dw + ((long)co * C + c) * 9 + tap, acc[tap]);
    if (c == 0) atomicAdd(db + co, bsum);
}

// ------------------------------------------------------------------------------------- weight re-layouts
// dst[i0][i1][i2][i3] (contiguous) <- src[i0*s0 + i1*s1 + i2*s2 + i3*s3];  ACC = false: dst bf16 = cast(src),
// ACC = true: dst fp32 += src   (GEMM-operand views of conv weights; weight gradients back into parameter layout)
template <bool ACC>
__global__ __launch_bounds__(256) void permute4_kernel(const float* __restrict__ src, long s0, long s1, long s2, long s3,
                                                       int n1, int n2, int n3, long total, void* __restrict__ dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int i3 = (int)(i % n3), i2 = (int)((i / n3) % n2), i1 = (int)((i / ((long)n3 * n2)) % n1);
    const long i0 = i / ((long)n3 * n2 * n1);
    const float v = src[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
    if (ACC) reinterpret_cast<float*>(dst)[i] += v;
    else reinterpret_cast<bf16_t*>(dst)[i] = f2bf(v);
}

}  // namespace ccd
