// cls_tail.h - the tail of the segmentation head (Dino/modules/segmentor.py:84-95: unpool2's BatchNorm2d + ReLU, then
// `cls` = Conv2d(128, 2, 3, padding=1)) as three kernels that stream the 268-MB map y (the transposed conv's output,
// bf16 [images*32*128, 128]) once or twice instead of eleven times.
//
// This level is HBM-bound and its neighbours' bytes are its whole cost: unfused (conv.h) the forward was bn_relu_fwd (read y,
// write a) + an NT GEMM (read a, write 18 fp32 tap planes) + a 9-point gather of the planes = 0.43 ms, the backward
// cls_grad_cols (write g [pixels, 64]) + dx GEMM (read g, write dx) + TN GEMM (read g, a) + colsum(g) + BN reduce (read dx, y)
// + BN apply (read dx, y, write dy) + colsum(dy) = 1.16 ms at 2 - 2.9 TB/s each.  Here:
//
//   cls_tail_fwd          a = relu(bn(y)) is formed in registers as the B operand of the tap-plane product
//                         z[k = class*9 + tap][pixel] = sum_c w[class][c][tap] a[pixel][c]  (16x16x32 MFMA, K = channels: the
//                         lane's 16-byte loads ARE the operand), a ring of four z rows in LDS, logits = bias + 9-point sum.
//                         `a` is never written; the backward recomputes it from y.
//   cls_tail_bwd<false>   d(a) = sum_k g[pixel][k] w[k][c], g[pixel][class*9 + tap] = dlogits[class][pixel - tap offset]
//                         (one MFMA per 16 pixels x 16 channels, K = 18 padded to 32), masked by a > 0, and the two sums
//                         BatchNorm's input gradient needs - nothing is written but those 256 floats (+ cls.bias's gradient).
//   cls_tail_bwd<true>    the same product again (cheaper than a 268-MB round trip), dy = gamma rstd (d - red0/n - xhat red1/n)
//                         written once, its column sums (the transposed conv's bias gradient) and cls.weight's gradient
//                         sum_pixels a[pixel][c] g[pixel][k] accumulated on the way: the contraction runs over the pixels, which
//                         both operands hold one-per-lane, so 32 pixels x 32 channels of `a` go through a wave-private LDS tile and
//                         come back transposed (8 consecutive pixels of one channel per lane), g is staged transposed once per band:
//                         4 MFMAs per 32 pixels and 16 accumulators (as 144 VALU accumulators per lane it held one wave per SIMD: 0.71 ms).
//
// Between the two backward kernels the host all-reduces the 256 sums under SyncBatchNorm, as between bn_relu_bwd_reduce / _apply.
// Layout contracts: C = 128 channels, 32 x 128 pixels per image, 2 classes, 3 x 3 taps (the reference's head; checked by the ABI,
// other shapes take the unfused kernels of conv.h).
#pragma once

namespace ccd {

constexpr int CT_C = 128, CT_H = 32, CT_W = 128, CT_WP = CT_W + 2, CT_K = 18;
constexpr int CT_GP = 40;                          // pitch of a gcol row in bf16 (32 + 8: 16 pixels x 16-byte reads touch 64 distinct banks)

// o = relu-argument of BatchNorm exactly as bn_relu_fwd_kernel forms it
__device__ __forceinline__ float ct_pre(float v, float mu, float rs, float ga, float be) { return (v - mu) * rs * ga + be; }

// ---------------------------------------------------------------------------------------------------------- forward
// grid = images * (32 / band_rows); 256 threads.  A wave owns two of a row's eight 16-pixel tiles and all 128 channels.
__global__ __launch_bounds__(256, 2) void cls_tail_fwd_kernel(const bf16_t* __restrict__ y, long ldy,
                                                           const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ logits,
                                                           int band_rows) {
    __shared__ float zr[4][CT_K][CT_WP];                                   // z rows (zi & 3), x + 1; columns 0 and W + 1 stay zero
    const int bands = CT_H / band_rows;
    const int n = blockIdx.x / bands, y0 = (blockIdx.x % bands) * band_rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, g = lane >> 4;
    for (int i = tid; i < 4 * CT_K * 2; i += 256) zr[i / (2 * CT_K)][(i >> 1) % CT_K][(i & 1) ? CT_WP - 1 : 0] = 0.f;
    // A operand: rows = tap planes k = 16 mt + (lane & 15), columns = the channels 32 ks + 8 g + e of k-step ks
    bf16x8 wa[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 16 * mt + px;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = k < CT_K ? w[((k / 9) * CT_C + 32 * ks + 8 * g + e) * 9 + k % 9] : 0.f;
            wa[mt][ks] = __builtin_bit_cast(bf16x8, pack8(v));
        }
    float mu[32], rs[32], ga[32], be[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int c = 32 * (i >> 3) + 8 * g + (i & 7);
        mu[i] = mean_rstd[c]; rs[i] = mean_rstd[CT_C + c]; ga[i] = gamma[c]; be[i] = beta[c];
    }
    // buffer addressing: the image's descriptor + a 32-bit lane offset (64-bit per-lane addresses of the tiles in flight cost a wave of occupancy)
    const unsigned ld_b = (unsigned)ldy * 2u;
    const buf_rsrc yr = make_rsrc(y + (long)n * CT_H * CT_W * ldy, (unsigned)(CT_H * CT_W) * ld_b);
    const unsigned yo = (unsigned)px * ld_b + 16u * (unsigned)g;
    auto load_tile = [&](int row, int tile, u32x4 (&v)[4]) __attribute__((always_inline)) {
        const unsigned base = (unsigned)(row * CT_W + tile * 16) * ld_b;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) v[ks] = buf_load16(yr, yo, base + 64u * (unsigned)ks);
    };
    auto tile_planes = [&](const u32x4 (&v)[4], int slot, int tile) __attribute__((always_inline)) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float x[8];
            unpack8(v[ks], x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float o = ct_pre(x[e], mu[8 * ks + e], rs[8 * ks + e], ga[8 * ks + e], be[8 * ks + e]);
                x[e] = o > 0.f ? o : 0.f;
            }
            const bf16x8 b = __builtin_bit_cast(bf16x8, pack8(x));
            acc0 = mfma_16x16x32_bf16(wa[0][ks], b, acc0);
            acc1 = mfma_16x16x32_bf16(wa[1][ks], b, acc1);
        }
        const int xo = tile * 16 + px + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) zr[slot][4 * g + r][xo] = acc0[r];                 // planes 0 .. 15
        if (g == 0) { zr[slot][16][xo] = acc1[0]; zr[slot][17][xo] = acc1[1]; }        // planes 16, 17
    };
    const int z_first = y0 - 1, z_last = y0 + band_rows;                               // z rows this band needs (clipped to the image below)
    u32x4 cur[4], nxt[4];
    {
        const int r0 = z_first < 0 ? 0 : z_first;
        load_tile(r0, 2 * wave, cur);
    }
    __syncthreads();                                                                   // (the zeroed padding columns)
    for (int zi = z_first; zi <= z_last; ++zi) {
        const int slot = zi & 3;
        if (zi >= 0 && zi < CT_H) {
            load_tile(zi, 2 * wave + 1, nxt);
            tile_planes(cur, slot, 2 * wave);
            const int zn = zi + 1;                                                     // next row's first tile, one tile ahead
            if (zn <= z_last && zn < CT_H) load_tile(zn, 2 * wave, cur);
            tile_planes(nxt, slot, 2 * wave + 1);
        }
        __syncthreads();
        const int o = zi - 1;
        if (o >= y0 && o < y0 + band_rows) {
            const int x = tid & 127, co = tid >> 7;
            float a = bias[co];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int sy = o + dy;
                if (sy >= 0 && sy < CT_H) {
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) a += zr[sy & 3][co * 9 + (dy + 1) * 3 + dx + 1][x + dx + 1];
                }
            }
            logits[(((long)n * 2 + co) * CT_H + o) * CT_W + x] = a;
        }
    }
}

// --------------------------------------------------------------------------------------------------------- backward
// grid-stride over bands of RB rows (RB * 8 pixel tiles); 256 threads.  A wave owns 32 channels (lane (pixel = lane & 15, g): channels 32 w + 8 g + e)
// and walks the band's 32 pixel tiles.  APPLY = false: red [2C] += the two sums, db_cls [2] += sum dlogits.
// APPLY = true: red holds the (rank-summed) sums; dy, dbias_t (column sums of the rounded dy), dw_cls [2][C][3][3] +=, and block 0
// adds this rank's own sums to dgamma / dbeta (as bn_relu_bwd_apply_kernel does).
template <bool APPLY, int RB>
__global__ __launch_bounds__(256, 2) void cls_tail_bwd_kernel(const float* __restrict__ dl, const bf16_t* __restrict__ y, long ldy,
                                                           const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ w,
                                                           float* __restrict__ red, float count, const float* __restrict__ red_local,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dw_cls, float* __restrict__ db_cls,
                                                           float* __restrict__ dbias_t, bf16_t* __restrict__ dy, long lddy,
                                                           int nbands) {
    __shared__ float dls[2][RB + 2][CT_WP];                               // dlogits rows y0 - 1 .. y0 + RB, x + 1, zero outside the image
    __shared__ __attribute__((aligned(16))) bf16_t gcol[RB * CT_W][CT_GP];  // g[pixel][k], k = class * 9 + tap (18 .. 31 zero)
    // APPLY: g transposed, [k][band pixel] (row 18 = zeros: what the lanes of the second column tile beyond k = 17 read), and per wave
    // a tile of `a` transposed, [channel][32 pixels]
    __shared__ __attribute__((aligned(16))) bf16_t gT[APPLY ? CT_K + 1 : 1][APPLY ? RB * CT_W + 8 : 8];
    __shared__ __attribute__((aligned(16))) bf16_t aT[APPLY ? 4 : 1][APPLY ? 32 : 1][CT_GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, g = lane >> 4;
    const int c0 = 32 * wave + 8 * g;                                        // the lane's 8 channels
    if (APPLY && blockIdx.x == 0 && tid < CT_C) {
        dbeta[tid] += red_local[tid];
        dgamma[tid] += red_local[CT_C + tid];
    }
    // A operand of d(a)[channel][pixel]: row m = lane & 15 of tile mt <-> channel 32 w + 8 (m >> 2) + 4 mt + (m & 3), so that a lane's
    // accumulators of the two tiles are its 8 consecutive channels (D row 4 g + r <-> e = 4 mt + r)
    bf16x8 wd[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int ch = 32 * wave + 8 * (px >> 2) + 4 * mt + (px & 3);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * g + e;
            v[e] = k < CT_K ? w[((k / 9) * CT_C + ch) * 9 + k % 9] : 0.f;
        }
        wd[mt] = __builtin_bit_cast(bf16x8, pack8(v));
    }
    float mu[8], rs[8], ga[8], be[8], k0[8], k1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mu[e] = mean_rstd[c0 + e]; rs[e] = mean_rstd[CT_C + c0 + e]; ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e];
        k0[e] = APPLY ? red[c0 + e] * (1.0f / count) : 0.f;
        k1[e] = APPLY ? red[CT_C + c0 + e] * (1.0f / count) : 0.f;
    }
    float s1[8], s2[8];                                                      // REDUCE: the two sums.  APPLY: s1 = column sums of dy
    f32x4 dwa[2][2];                                 // APPLY: d cls.weight^T [channel 32 w + 16 mt + 4 g + r][k = 16 nt + (lane & 15)]
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) dwa[i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (APPLY) {
        for (int i = tid; i < RB * CT_W + 8; i += 256) gT[CT_K][i] = 0;
    }
    float db0 = 0.f, db1 = 0.f;
    constexpr int BPI = CT_H / RB, NJ = RB + 2, TILES = RB * CT_W / 16;
    for (int i = tid; i < 2 * NJ * 2; i += 256) dls[i / (2 * NJ)][(i >> 1) % NJ][(i & 1) ? CT_WP - 1 : 0] = 0.f;     // the padding columns, once
    // The staging of a band - the dlogits rows it needs, re-laid as the two products' B operands - stands between barriers: its loads
    // are issued a band AHEAD (NJ values per thread: 256 threads = two rows of 128) and spend the main loop in flight
    const int srow = tid >> 7, sx = tid & 127;
    float dlv[NJ];
    auto fetch_dl = [&](int band) __attribute__((always_inline)) {
        const int n = band / BPI, y0 = (band % BPI) * RB;
        const float* dimg = dl + (long)n * 2 * CT_H * CT_W + sx;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ri = 2 * j + srow, co = ri >= NJ ? 1 : 0, sy = y0 - 1 + ri - NJ * co;      // rows 0 .. NJ-1 of class 0, then class 1
            dlv[j] = (sy >= 0 && sy < CT_H) ? dimg[(co * CT_H + sy) * CT_W] : 0.f;
        }
    };
    // gcol: a thread builds the SAME 8-column chunk (tid & 3) of every pixel it visits: its 8 source offsets are fixed
    int goff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * (tid & 3) + e, co = k >= 9 ? 1 : 0, tap = k - 9 * co;
        goff[e] = k < CT_K ? (co * NJ + 1 - (tap / 3 - 1)) * CT_WP + 1 - (tap % 3 - 1) : -1;
    }
    const float* dflat = &dls[0][0][0];
    int band = blockIdx.x;
    if (band < nbands) fetch_dl(band);
    for (; band < nbands; band += gridDim.x) {
        const int n = band / BPI, y0 = (band % BPI) * RB;
        __syncthreads();                                                     // the previous band's readers are done
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ri = 2 * j + srow, co = ri >= NJ ? 1 : 0, r = ri - NJ * co;
            dls[co][r][sx + 1] = dlv[j];
            if (!APPLY && r >= 1 && r <= RB) { if (co == 0) db0 += dlv[j]; else db1 += dlv[j]; }
        }
        __syncthreads();
        // (buffer addressing: a wave-uniform descriptor of the band + 32-bit lane offsets - as 64-bit per-lane addresses the unrolled
        // tiles' pointers alone were 60 registers and a wave of occupancy)
        const unsigned ld_b = (unsigned)ldy * 2u, ldd_b = (unsigned)lddy * 2u;
        const buf_rsrc yr = make_rsrc(y + ((long)n * CT_H + y0) * CT_W * ldy, (unsigned)(RB * CT_W) * ld_b);
        const buf_rsrc dyr = make_rsrc(APPLY ? (const void*)(dy + ((long)n * CT_H + y0) * CT_W * lddy) : (const void*)y,
                                       APPLY ? (unsigned)(RB * CT_W) * ldd_b : 0u);
        const unsigned yo = (unsigned)px * ld_b + (unsigned)c0 * 2u, dyo = (unsigned)px * ldd_b + (unsigned)c0 * 2u;
        auto load4 = [&](u32x4 (&v)[4], int t0) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = buf_load16(yr, yo, (unsigned)((t0 + u) * 16) * ld_b);
        };
        u32x4 ya[4], yc[4];
        load4(ya, 0);                                                        // the first tiles and the next band's rows: in flight across the build
        if (band + (int)gridDim.x < nbands) fetch_dl(band + (int)gridDim.x);
#pragma unroll
        for (int j = 0; j < 2 * RB; ++j) {                                   // (pixel, chunk): g[q][class*9 + tap] = dl[class][q - offset(tap)]
            const int q = 64 * j + (tid >> 2), pbase = (j >> 1) * CT_WP + 64 * (j & 1) + (tid >> 2);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = goff[e] >= 0 ? dflat[pbase + goff[e]] : 0.f;
            *reinterpret_cast<u32x4*>(&gcol[q][8 * (tid & 3)]) = pack8(v);
        }
        if (APPLY) {
            for (int i = tid; i < CT_K * (RB * CT_W / 8); i += 256) {       // (k, 8 consecutive pixels of a row)
                const int k = i / (RB * CT_W / 8), q0 = (i % (RB * CT_W / 8)) * 8, r = q0 / CT_W, xx = q0 % CT_W;
                const int co = k >= 9 ? 1 : 0, tap = k - 9 * co;
                const float* src = &dls[co][r + 1 - (tap / 3 - 1)][xx + 1 - (tap % 3 - 1)];
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = src[e];
                *reinterpret_cast<u32x4*>(&gT[k][q0]) = pack8(v);
            }
        }
        __syncthreads();
        // four tiles in flight per lane beside the four being worked on (with 144 accumulators a SIMD holds ONE wave of the APPLY
        // kernel: the loads in flight are the lane's own)
        auto tile = [&](const u32x4& yw, int t, bool odd) __attribute__((always_inline)) {
            const int q = t * 16 + px;
            const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&gcol[q][8 * g]));
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 d0 = mfma_16x16x32_bf16(wd[0], b, zero), d1 = mfma_16x16x32_bf16(wd[1], b, zero);
            float xv[8], o[8], av[8];
            unpack8(yw, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                const float pre = xh * ga[e] + be[e];
                const float dd = e < 4 ? d0[e & 3] : d1[e & 3];
                const float d = pre > 0.f ? dd : 0.f;
                if (APPLY) {
                    o[e] = ga[e] * rs[e] * (d - k0[e] - xh * k1[e]);
                    av[e] = pre > 0.f ? bf2f(f2bf(pre)) : 0.f;                // the operand the forward product saw
                } else {
                    s1[e] += d;
                    s2[e] += d * xh;
                }
            }
            if (APPLY) {
                const u32x4 ow = pack8(o);
                buf_store16(dyr, dyo, (unsigned)(t * 16) * ldd_b, ow);
                float orr[8];
                unpack8(ow, orr);
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[e] += orr[e];
                // `a` of this tile, transposed into the wave's LDS tile: aT[channel 8 g + e][pixel (t & 1) * 16 + px]
                const u32x4 aw = pack8(av);
                const int pcol = (odd ? 16 : 0) + px;
                aT[wave][8 * g + 0][pcol] = (bf16_t)(aw.x & 0xffffu); aT[wave][8 * g + 1][pcol] = (bf16_t)(aw.x >> 16);
                aT[wave][8 * g + 2][pcol] = (bf16_t)(aw.y & 0xffffu); aT[wave][8 * g + 3][pcol] = (bf16_t)(aw.y >> 16);
                aT[wave][8 * g + 4][pcol] = (bf16_t)(aw.z & 0xffffu); aT[wave][8 * g + 5][pcol] = (bf16_t)(aw.z >> 16);
                aT[wave][8 * g + 6][pcol] = (bf16_t)(aw.w & 0xffffu); aT[wave][8 * g + 7][pcol] = (bf16_t)(aw.w >> 16);
                if (odd) {                                                    // 32 pixels are in: dW^T += a^T g (K = the 32 pixels)
                    wave_lds_fence();
                    const int q32 = (t - 1) * 16 + 8 * g;                     // the lane's 8 pixels of the pair
                    bf16x8 af[2], gf[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) af[mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&aT[wave][16 * mt + px][8 * g]));
                    gf[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&gT[px][q32]));
                    gf[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&gT[px < 2 ? 16 + px : CT_K][q32]));
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) dwa[mt][nt] = mfma_16x16x32_bf16(af[mt], gf[nt], dwa[mt][nt]);
                    wave_lds_fence();                                         // (the next pair's writes come after these reads)
                }
            }
        };
#pragma unroll 1
        for (int t0 = 0; t0 < TILES; t0 += 8) {
            load4(yc, t0 + 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) { tile(ya[u], t0 + u, u & 1); CCD_SCHED_FENCE(); }       // (tiles one after the other: interleaved, their temporaries cost a wave of occupancy)
            load4(ya, t0 + 8);                 // (unconditional: past the band's end the descriptor returns zeros without a memory access - behind a
                                               //  branch the compiler loses count of what is in flight and drains the queue, vmcnt(0), every 8 tiles)
#pragma unroll
            for (int u = 0; u < 4; ++u) { tile(yc[u], t0 + 4 + u, u & 1); CCD_SCHED_FENCE(); }
        }
    }
    // totals over the 16 pixel lanes, then one atomic per (lane group, value)
    auto over_pixels = [&](float v) __attribute__((always_inline)) {
        v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4); v += shfl_xor(v, 8);
        return v;
    };
    if (!APPLY) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = over_pixels(s1[e]), b = over_pixels(s2[e]);
            if (px == 0) { atomicAdd(red + c0 + e, a); atomicAdd(red + CT_C + c0 + e, b); }
        }
        db0 = wave_sum(db0); db1 = wave_sum(db1);
        if (lane == 0) { atomicAdd(db_cls, db0); atomicAdd(db_cls + 1, db1); }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = over_pixels(s1[e]);
            if (px == 0) atomicAdd(dbias_t + c0 + e, a);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int k = 16 * nt + px;
                if (k < CT_K) {
                    const int co = k / 9, tap = k % 9;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        atomicAdd(dw_cls + ((long)co * CT_C + 32 * wave + 16 * mt + 4 * g + r) * 9 + tap, dwa[mt][nt][r]);
                }
            }
    }
}

}  // namespace ccd
