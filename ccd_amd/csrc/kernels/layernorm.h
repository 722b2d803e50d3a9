// layernorm.h - LayerNorm(eps) over the fp32 residual stream, bf16 out (vision_transformer.py:99,103,156,162).
// One wave per token row, 16-byte fp32 / 8-byte bf16 accesses (4 elements per lane and step; STEPS = 2 for E <= 512, 4 for E <= 1024), statistics
// by wave shuffles.  HBM-bound: algorithmic bytes per row = 4E (x) + 2E (y) forward;
// backward 2E (dy) + 4E (x) + 8E (g read+write) [+ 2E for the fused bf16 copy of the updated gradient stream].
#pragma once

namespace ccd {

constexpr int LN_VEC = 4;                        // lane l owns elements 4*(l + 64*s) .. +3, s < LN_STEPS (template parameter)
__host__ __device__ constexpr int ln_steps(int E) { return E <= 512 ? 2 : 4; }       // E <= 1024 (vit_base_768: 3 live steps of 4)

template <int LN_STEPS>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int rows, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                       // whole waves exit together; no block barrier below
    const float* xr = x + (long)row * E;
    f32x4v v[LN_STEPS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_STEPS; ++i) {
        const int c = LN_VEC * (lane + 64 * i);
        f32x4v t = {0.f, 0.f, 0.f, 0.f};
        if (c < E) t = *reinterpret_cast<const f32x4v*>(xr + c);
        v[i] = t;
        s += (t.x + t.y) + (t.z + t.w);
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_STEPS; ++i) {
        const int c = LN_VEC * (lane + 64 * i);
        if (c < E) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
    bf16_t* yr = y + (long)row * E;
#pragma unroll
    for (int i = 0; i < LN_STEPS; ++i) {
        const int c = LN_VEC * (lane + 64 * i);
        if (c < E) {
            const f32x4v g = *reinterpret_cast<const f32x4v*>(gamma + c);
            const f32x4v b = *reinterpret_cast<const f32x4v*>(beta + c);
            u32x2 o;
            o.x = pack_bf2((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y);
            o.y = pack_bf2((v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
            *reinterpret_cast<u32x2*>(yr + c) = o;
        }
    }
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

// g[row,:] (+)= LN_backward(dy[row,:]) ; dgamma += sum_rows dy*xhat ; dbeta += sum_rows dy   (fp32 atomics)
//   ACCUM = true : g += dx   (residual-gradient stream of a transformer block)
//   ACCUM = false: g  = dx   (final norm: starts the stream)
// Optional fused tail (gb != null): gb[row,:] = bf16(g_new[row,:] * rowscale[row / rows_per_sample]) - the gradient
// entering the NEXT residual branch with that branch's DropPath scale - and dbias += column sums of gb (the bias
// gradient of that branch's output projection).  Saves a pass over g and a pass over gb per branch.
// G16 (round 6): the gradient stream g is bf16 (read, accumulated in fp32, rounded once per writer).
template <bool ACCUM, int LN_STEPS, bool G16 = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, void* __restrict__ g_,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     bf16_t* __restrict__ gb, const float* __restrict__ rowscale,
                                                     int rows_per_sample, float* __restrict__ dbias, int rows, int E,
                                                     int rows_per_block) {
    // column sums of the four waves meet in 6 KiB of LDS by ds_add_f32 (every lane its own columns: conflict-free) - small
    // enough that two blocks fit NEXT to a 144-KiB weight-gradient workgroup on the same CU (engine.py runs this HBM-bound
    // kernel beside the MFMA-bound gemm_tn384 kernel of the other stream when the LayerNorm backward is not fused)
    __shared__ float red[3][64 * LN_VEC * LN_STEPS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < 3 * 64 * LN_VEC * LN_STEPS; c += 256) (&red[0][0])[c] = 0.f;
    __syncthreads();
    const int row_begin = blockIdx.x * rows_per_block;
    const int row_end = row_begin + rows_per_block < rows ? row_begin + rows_per_block : rows;
    f32x4v gam[LN_STEPS], dg[LN_STEPS], db[LN_STEPS], dbi[LN_STEPS];
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < LN_STEPS; ++i) {
        const int c = LN_VEC * (lane + 64 * i);
        gam[i] = c < E ? *reinterpret_cast<const f32x4v*>(gamma + c) : zero;
        dg[i] = zero;
        db[i] = zero;
        dbi[i] = zero;
    }
    for (int row = row_begin + w; row < row_end; row += 4) {
        // everything the row needs is requested up front - x, dy AND the gradient row it accumulates into (requested behind the
        // two wave reductions it used to wait for a second HBM round trip per row) - and has arrived before the first store
        const float* xr = x + (long)row * E;
        const bf16_t* dyr = dy + (long)row * E;
        float* gr = reinterpret_cast<float*>(g_) + (long)row * E;
        bf16_t* gr16 = reinterpret_cast<bf16_t*>(g_) + (long)row * E;
        f32x4v xv[LN_STEPS], gv[LN_STEPS];
        u32x2 dw[LN_STEPS];
#pragma unroll
        for (int i = 0; i < LN_STEPS; ++i) {
            const int c = LN_VEC * (lane + 64 * i);
            xv[i] = zero; gv[i] = zero; dw[i] = u32x2{0u, 0u};
            if (c < E) {
                xv[i] = *reinterpret_cast<const f32x4v*>(xr + c);
                dw[i] = *reinterpret_cast<const u32x2*>(dyr + c);
                if (ACCUM) {
                    if constexpr (G16) {
                        const u32x2 h = *reinterpret_cast<const u32x2*>(gr16 + c);
                        gv[i] = f32x4v{bf_lo(h.x), bf_hi(h.x), bf_lo(h.y), bf_hi(h.y)};
                    } else gv[i] = *reinterpret_cast<const f32x4v*>(gr + c);
                }
            }
        }
        const float mu = mean[row], rs = rstd[row];
        const float sc = (gb && rowscale) ? rowscale[row / rows_per_sample] : 1.0f;
        f32x4v xh[LN_STEPS], d[LN_STEPS];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_STEPS; ++i) {
            const int c = LN_VEC * (lane + 64 * i);
            xh[i] = zero;
            d[i] = zero;
            if (c < E) {
                const f32x4v dyv = {bf_lo(dw[i].x), bf_hi(dw[i].x), bf_lo(dw[i].y), bf_hi(dw[i].y)};
                xh[i] = (xv[i] - mu) * rs;
                d[i] = dyv * gam[i];
                dg[i] += dyv * xh[i];
                db[i] += dyv;
                s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
                const f32x4v e = d[i] * xh[i];
                s2 += (e.x + e.y) + (e.z + e.w);
            }
        }
        s1 = wave_sum(s1) / (float)E;
        s2 = wave_sum(s2) / (float)E;
        if (ACCUM) {
#pragma unroll
            for (int i = 0; i < LN_STEPS; ++i) needed_here(gv[i]);
        }
#pragma unroll
        for (int i = 0; i < LN_STEPS; ++i) {
            const int c = LN_VEC * (lane + 64 * i);
            if (c < E) {
                f32x4v dx = (d[i] - s1 - xh[i] * s2) * rs;
                if (ACCUM) dx += gv[i];
                if constexpr (G16) {
                    u32x2 h;
                    h.x = pack_bf2(dx.x, dx.y);
                    h.y = pack_bf2(dx.z, dx.w);
                    *reinterpret_cast<u32x2*>(gr16 + c) = h;
                } else *reinterpret_cast<f32x4v*>(gr + c) = dx;
                if (gb) {
                    const f32x4v o = dx * sc;
                    u32x2 pk;
                    pk.x = pack_bf2(o.x, o.y);
                    pk.y = pack_bf2(o.z, o.w);
                    *reinterpret_cast<u32x2*>(gb + (long)row * E + c) = pk;
                    // sum what the GEMMs will actually read (the bf16-rounded values)
                    const f32x4v rq = {bf_lo(pk.x), bf_hi(pk.x), bf_lo(pk.y), bf_hi(pk.y)};
                    dbi[i] += rq;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_STEPS; ++i) {
        const int c = LN_VEC * (lane + 64 * i);
        const float vg[4] = {dg[i].x, dg[i].y, dg[i].z, dg[i].w}, vb[4] = {db[i].x, db[i].y, db[i].z, db[i].w};
        const float vi[4] = {dbi[i].x, dbi[i].y, dbi[i].z, dbi[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&red[0][c + e], vg[e]);
            atomicAdd(&red[1][c + e], vb[e]);
            if (gb && dbias) atomicAdd(&red[2][c + e], vi[e]);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) {
        atomicAdd(dgamma + c, red[0][c]);
        atomicAdd(dbeta + c, red[1][c]);
        if (gb && dbias) atomicAdd(dbias + c, red[2][c]);
    }
}

}  // namespace ccd
