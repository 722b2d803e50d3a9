// attention_bwd.h - backward of the T=256, d=64 attention of attention_fwd.h, as two kernels per layer:
//   attention_bwd_dq_kernel   one WG (8 waves x 32 queries) per (view, head): D = rowsum(dO*O), dQ
//   attention_bwd_dkv_kernel  one WG (8 waves x 32 keys)    per (view, head): dK, dV   (both persistent: one WG per CU)
// P is recomputed from the saved LSE (never stored).  Both kernels use the transposed-product trick of the
// forward: the probability / dS tile that comes out of one MFMA is directly the B operand of the next one,
// and the LDS-resident A operands (K^T, Q^T, dO^T images) carry the matching key/query permutation.
//   dP = dO V^T ; dS = P * (dP - D) * scale ; dQ = dS K ; dK = dS^T Q ; dV = P^T dO
#pragma once

namespace ccd {

constexpr int ATTB_THREADS = 512;
constexpr int ATTB_IMG = ATT_T * ATT_D * 2;                    // 32 KiB per operand image
constexpr int ATTB_MAX_HEADS = 16;
constexpr int ATTB_CS_BYTES = ATTB_MAX_HEADS * ATT_D * 4;      // per-head column sums of one gradient (qkv-bias gradient)
constexpr int ATTB_DQ_SMEM = 3 * ATTB_IMG + ATTB_CS_BYTES;     // K, V, K^T, column sums of dQ
constexpr int ATTB_DKV_SMEM = 4 * ATTB_IMG + 2 * ATT_T * 4;    // Q, dO, Q^T, dO^T, lse, D

// The qkv-bias gradient (column sums of d_qkv over all token rows; vision_transformer.py:75 `qkv_bias`, autograd's
// grad_output.sum(0) of the qkv Linear) without a pass over d_qkv (round 2: 12 colsum_bf16 launches over 302 MB each):
//   k part: sum_key dK[key][:] = sum_q (sum_key dS[q][key]) Q[q][:] and sum_key dS[q][key] = scale (sum_key P dP - D sum_key P)
//           = scale (D - D) = 0: the softmax is shift invariant, the key bias has NO gradient.  Nothing is added (the reference's
//           fp32 sum is rounding residue of the order 1e-7 there).
//   v part: sum_key dV[key][:] = sum_q (sum_key P[q][key]) dO[q][:] = column sums of dO, which the caller knows without
//           touching dO: dO = gb . Wproj, so colsum(dO) = colsum(gb) . Wproj = (proj.bias gradient) . Wproj - a 384 x 384 matvec.
//   q part: has no closed form: the dQ kernel sums its fp32 result tiles over their 32 rows with the DPP halving tree of
//           rowgemm.h (rg_colsum16), adds them to a per-head LDS accumulator of the (persistent) workgroup and writes ONE row of
//           partial sums bias_ws[workgroup][E] at its end (plain stores: same-address fp32 atomics from 256 workgroups cost more
//           than the pass they replace).
// qkv_bias_finish_kernel adds the <= 256 partial rows up and forms the matvec.  (Fusing the k / v sums into the dK / dV kernel
// as well was measured first: + 69 us per layer for both kernels - that kernel already spills at 256 registers - against 56 us
// for the separate pass.)
__device__ __forceinline__ void attb_colsum_tiles(const f32x16 (&acc)[2], float* cs_head, int lq, int hf) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[dt][r];
        rg_colsum16(v, cs_head + 32 * dt, lq, hf);
    }
}
// blockIdx.y = 0: d_bias[c] += sum_r ws[r][c] (r < rows: the dQ kernel's partial rows), c < E
// blockIdx.y = 1: d_bias[2 E + c] += vec[c] (mat == null) or sum_i vec[i] mat[i][c] (colsum(dO) = proj.bias gradient . Wproj)
// 64 columns x 16 row groups per workgroup
__global__ __launch_bounds__(1024) void qkv_bias_finish_kernel(const float* __restrict__ ws, int rows, const float* __restrict__ vec,
                                                              const float* __restrict__ mat, long ld_mat, int E,
                                                              float* __restrict__ d_bias) {
    __shared__ float part[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float a = 0.f;
    if (c < E) {
        // (eight loads in flight per thread: one load per trip made the 16 - 24 trips a chain of L2 round trips, 13 us for 1 MB)
        if (blockIdx.y == 0) {
            int r = rg;
            for (; r + 7 * 16 < rows; r += 8 * 16) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ws[(long)(r + 16 * u) * E + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            for (; r < rows; r += 16) a += ws[(long)r * E + c];
        } else if (mat) {
            int i = rg;
            for (; i + 7 * 16 < E; i += 8 * 16) {
                float x[8], m[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { x[u] = vec[i + 16 * u]; m[u] = mat[(long)(i + 16 * u) * ld_mat + c]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) a = fmaf(x[u], m[u], a);
            }
            for (; i < E; i += 16) a = fmaf(vec[i], mat[(long)i * ld_mat + c], a);
        } else if (rg == 0) {
            a = vec[c];
        }
    }
    part[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && c < E) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][cl];
        d_bias[(blockIdx.y ? 2 * E : 0) + c] += t;
    }
}
__device__ __forceinline__ void attb_stage_rows(const bf16_t* __restrict__ src, long row_stride, char* img) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = t + ATTB_THREADS * i, row = id >> 3, slot = id & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + (long)row * row_stride + slot * 8);
        *reinterpret_cast<u32x4*>(img + row * 128 + ((slot ^ ((row >> 1) & 7)) * 16)) = v;
    }
}
__device__ __forceinline__ void attb_stage_transposed(const bf16_t* __restrict__ src, long row_stride, char* img) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kb = 16 * (w & 3) + (lane & 15);
    const int db = (lane >> 4) + 4 * (w >> 2);
    u32x4 r[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
        r[kq] = *reinterpret_cast<const u32x4*>(src + (long)(4 * kb + kq) * row_stride + db * 8);
    const int chunk = 4 * (kb >> 2) + att_chunk_pos(kb & 3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = 8 * db + j;
        const unsigned w0 = r[0][j >> 1], w1 = r[1][j >> 1], w2 = r[2][j >> 1], w3 = r[3][j >> 1];
        u32x2 o;
        o.x = perm_b32(w1, w0, (j & 1) ? 0x07060302u : 0x05040100u);
        o.y = perm_b32(w3, w2, (j & 1) ? 0x07060302u : 0x05040100u);
        const int slot = (chunk >> 1) ^ (d & 15);
        *reinterpret_cast<u32x2*>(img + d * 512 + slot * 16 + (chunk & 1) * 8) = o;
    }
}
__device__ __forceinline__ bf16x8 attb_row_frag(const char* img, int row, int kk, int hf) {
    return *reinterpret_cast<const bf16x8*>(img + row * 128 + (((2 * kk + hf) ^ ((row >> 1) & 7)) * 16));
}
__device__ __forceinline__ bf16x8 attb_tr_frag(const char* img, int d, int ks, int hf) {
    return *reinterpret_cast<const bf16x8*>(img + d * 512 + (((2 * ks + hf) ^ (d & 15)) * 16));
}
__device__ __forceinline__ void attb_store_t(bf16_t* row_ptr, const f32x16 (&acc)[2], int hf, bool live = true) {
    att_store_row16(row_ptr, acc, hf, 1.0f, live);           // (attention_fwd.h: 16-byte stores after a half-wave exchange)
}

// Register images of one (view, head) block's operands: what a thread moves to LDS / consumes itself.  The kernels are
// persistent (one workgroup per CU walks the blocks) and request block b+1's pieces right after block b's have been
// written to LDS, so the HBM / L2 latency of the staging is spent under block b's MFMAs instead of in front of them
// (measured before: 16 us per block for ~5 us of MFMA + softmax work).
struct AttbRows { u32x4 r[4]; };
__device__ __forceinline__ void attb_load_rows(const bf16_t* __restrict__ src, long row_stride, AttbRows& x) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = t + ATTB_THREADS * i, row = id >> 3, slot = id & 7;
        x.r[i] = *reinterpret_cast<const u32x4*>(src + (long)row * row_stride + slot * 8);
    }
}
__device__ __forceinline__ void attb_store_rows(const AttbRows& x, char* img) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = t + ATTB_THREADS * i, row = id >> 3, slot = id & 7;
        *reinterpret_cast<u32x4*>(img + row * 128 + ((slot ^ ((row >> 1) & 7)) * 16)) = x.r[i];
    }
}
__device__ __forceinline__ void attb_load_transposed(const bf16_t* __restrict__ src, long row_stride, AttbRows& x) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kb = 16 * (w & 3) + (lane & 15);
    const int db = (lane >> 4) + 4 * (w >> 2);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
        x.r[kq] = *reinterpret_cast<const u32x4*>(src + (long)(4 * kb + kq) * row_stride + db * 8);
}
__device__ __forceinline__ void attb_store_transposed(const AttbRows& x, char* img) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kb = 16 * (w & 3) + (lane & 15);
    const int db = (lane >> 4) + 4 * (w >> 2);
    const int chunk = 4 * (kb >> 2) + att_chunk_pos(kb & 3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = 8 * db + j;
        const unsigned w0 = x.r[0][j >> 1], w1 = x.r[1][j >> 1], w2 = x.r[2][j >> 1], w3 = x.r[3][j >> 1];
        u32x2 o;
        o.x = perm_b32(w1, w0, (j & 1) ? 0x07060302u : 0x05040100u);
        o.y = perm_b32(w3, w2, (j & 1) ? 0x07060302u : 0x05040100u);
        const int slot = (chunk >> 1) ^ (d & 15);
        *reinterpret_cast<u32x2*>(img + d * 512 + slot * 16 + (chunk & 1) * 8) = o;
    }
}

// the row image written from the registers of the TRANSPOSED loader (a thread holds rows 4 kb .. 4 kb + 3, columns 8 db .. + 7):
// one set of loads feeds both images of an operand
__device__ __forceinline__ void attb_store_rows_of_transposed(const AttbRows& x, char* img) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kb = 16 * (w & 3) + (lane & 15);
    const int db = (lane >> 4) + 4 * (w >> 2);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        const int row = 4 * kb + kq;
        *reinterpret_cast<u32x4*>(img + row * 128 + ((db ^ ((row >> 1) & 7)) * 16)) = x.r[kq];
    }
}

// The two waves of a SIMD (w and w + 4) run the same MFMA / softmax / MFMA phases in lock step after every block barrier -
// the matrix pipe, the VALU and the LDS take turns instead of overlapping.  Holding waves 4..7 back by about one MFMA phase
// (skew x 64 cycles) makes the pairs alternate.
__device__ __forceinline__ void attb_skew(int w, int skew) {
    if (w >= 4) wave_nap(skew);
}

__global__ __launch_bounds__(512) void attention_bwd_dq_kernel(const bf16_t* __restrict__ qkv,
                                                               const bf16_t* __restrict__ o,
                                                               const bf16_t* __restrict__ d_o,
                                                               const float* __restrict__ lse,
                                                               float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                               float* __restrict__ bias_ws,
                                                               int heads, float scale, int nblocks, int skew) {
    char* smem = dynamic_smem();
    char* k_img = smem;
    char* v_img = smem + ATTB_IMG;
    char* kt_img = smem + 2 * ATTB_IMG;
    float* cs = reinterpret_cast<float*>(smem + 3 * ATTB_IMG);       // [heads][64] column sums of dQ over this workgroup's blocks
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int E = heads * ATT_D;
    if (bias_ws)
        for (int i = threadIdx.x; i < E; i += ATTB_THREADS) cs[i] = 0.f;     // (published by the first block's barrier)
    const long rs3 = 3L * E;
    const int q = 32 * w + lq;
    AttbRows vr, ktr;                                         // (K: one load set feeds the row and the transposed image)
    u32x4 qw[4], dw[4], ow[4];
    float lse_n = 0.f;
    auto request = [&](int blk) {                             // everything this thread needs of block `blk`
        const int view = blk / heads, head = blk % heads;
        const bf16_t* q_base = qkv + (long)view * ATT_T * rs3 + head * ATT_D;
        attb_load_rows(q_base + 2 * E, rs3, vr);
        attb_load_transposed(q_base + E, rs3, ktr);
        const long orow = ((long)view * ATT_T + q) * E + head * ATT_D;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qw[kk] = *reinterpret_cast<const u32x4*>(q_base + (long)q * rs3 + 16 * kk + 8 * hf);
            dw[kk] = *reinterpret_cast<const u32x4*>(d_o + orow + 16 * kk + 8 * hf);
            ow[kk] = *reinterpret_cast<const u32x4*>(o + orow + 16 * kk + 8 * hf);
        }
        lse_n = lse[((long)view * heads + head) * ATT_T + q];
    };
    int blk = blockIdx.x;
    if (blk < nblocks) request(blk);
    for (; blk < nblocks; blk += gridDim.x) {
        const int view = blk / heads, head = blk % heads;
        attb_store_rows_of_transposed(ktr, k_img);
        attb_store_rows(vr, v_img);
        attb_store_transposed(ktr, kt_img);
        bf16x8 qf[4], dof[4];
        float dsum = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[kk] = __builtin_bit_cast(bf16x8, qw[kk]);
            dof[kk] = __builtin_bit_cast(bf16x8, dw[kk]);
            float a[8], b[8];
            unpack8(dw[kk], a);
            unpack8(ow[kk], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += a[e] * b[e];
        }
        dsum += shfl_xor(dsum, 32);
        const long stat = ((long)view * heads + head) * ATT_T + q;
        const float my_lse = lse_n;
        if (hf == 0) delta[stat] = dsum;
        __syncthreads();
        if (blk + (int)gridDim.x < nblocks) request(blk + gridDim.x);      // flies under this block's MFMAs
        attb_skew(w, skew);

        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < 8; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            const int row = 32 * kt + lq;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = mfma_32x32x16_bf16(attb_row_frag(k_img, row, kk, hf), qf[kk], s);
                dp = mfma_32x32x16_bf16(attb_row_frag(v_img, row, kk, hf), dof[kk], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(fmaf(s[r], scale * 1.4426950408889634f, -my_lse * 1.4426950408889634f));
                s[r] = p * (dp[r] - dsum) * scale;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 dsf;
#pragma unroll
                for (int e = 0; e < 8; ++e) dsf[e] = (short)f2bf(s[8 * s2 + e]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = mfma_32x32x16_bf16(attb_tr_frag(kt_img, 32 * dt + lq, 2 * kt + s2, hf), dsf, dq[dt]);
            }
        }
        attb_store_t(dqkv + ((long)view * ATT_T + q) * rs3 + head * ATT_D, dq, hf);
        if (bias_ws) attb_colsum_tiles(dq, cs + head * ATT_D, lq, hf);
        __syncthreads();                                     // the images are rewritten by the next block
    }
    if (bias_ws)                                             // (the loop's last barrier published every wave's sums)
        for (int i = threadIdx.x; i < E; i += ATTB_THREADS) bias_ws[(long)blockIdx.x * E + i] = cs[i];
}

__global__ __launch_bounds__(512) void attention_bwd_dkv_kernel(const bf16_t* __restrict__ qkv,
                                                                const bf16_t* __restrict__ d_o,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dqkv, int heads, float scale, int nblocks,
                                                                int skew) {
    char* smem = dynamic_smem();
    char* q_img = smem;
    char* do_img = smem + ATTB_IMG;
    char* qt_img = smem + 2 * ATTB_IMG;
    char* dot_img = smem + 3 * ATTB_IMG;
    float* lse_s = reinterpret_cast<float*>(smem + 4 * ATTB_IMG);
    float* del_s = lse_s + ATT_T;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int E = heads * ATT_D;
    const long rs3 = 3L * E;
    const int key = 32 * w + lq;
    // persistent like the dQ kernel: everything a thread needs of block b + gridDim.x is requested right after block b's
    // pieces have been written to LDS (measured: 214 -> ... us per launch; the staging latency of 128 KiB per block used to
    // sit in front of every block's ~5 us of MFMA work)
    AttbRows qtr, dotr;                                       // (one load set per operand feeds its row AND its transposed image)
    u32x4 kw[4], vw[4];
    float lse_r = 0.f, del_r = 0.f;
    auto request = [&](int blk) {
        const int view = blk / heads, head = blk % heads;
        const bf16_t* q_base = qkv + (long)view * ATT_T * rs3 + head * ATT_D;
        const bf16_t* do_base = d_o + (long)view * ATT_T * E + head * ATT_D;
        attb_load_transposed(q_base, rs3, qtr);
        attb_load_transposed(do_base, E, dotr);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kw[kk] = *reinterpret_cast<const u32x4*>(q_base + E + (long)key * rs3 + 16 * kk + 8 * hf);
            vw[kk] = *reinterpret_cast<const u32x4*>(q_base + 2 * E + (long)key * rs3 + 16 * kk + 8 * hf);
        }
        if (threadIdx.x < ATT_T) {
            const long stat = ((long)view * heads + head) * ATT_T + threadIdx.x;
            lse_r = lse[stat];
            del_r = delta[stat];
        }
    };
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int view = blk / heads, head = blk % heads;
        request(blk);                                        // (no cross-block prefetch: its 64 registers are worth more as the
                                                             // second score tile of the software pipeline below)
        attb_store_rows_of_transposed(qtr, q_img);
        attb_store_rows_of_transposed(dotr, do_img);
        attb_store_transposed(qtr, qt_img);
        attb_store_transposed(dotr, dot_img);
        if (threadIdx.x < ATT_T) { lse_s[threadIdx.x] = lse_r; del_s[threadIdx.x] = del_r; }
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kf[kk] = __builtin_bit_cast(bf16x8, kw[kk]);
            vf[kk] = __builtin_bit_cast(bf16x8, vw[kk]);
        }
        __syncthreads();
        attb_skew(w, skew);

        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
        // software pipeline: the score products of tile qt + 1 are issued BEFORE the softmax arithmetic of tile qt, so the
        // matrix pipe works under the VALU block instead of waiting for it
        auto scores = [&](int qt, f32x16& s, f32x16& dp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            const int row = 32 * qt + lq;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = mfma_32x32x16_bf16(attb_row_frag(q_img, row, kk, hf), kf[kk], s);      // S[q][key]
                dp = mfma_32x32x16_bf16(attb_row_frag(do_img, row, kk, hf), vf[kk], dp);   // dP[q][key]
            }
        };
        f32x16 s, dp;
        scores(0, s, dp);
#pragma unroll 1
        for (int qt = 0; qt < 8; ++qt) {
            f32x16 sn, dpn;
            scores(qt + 1 < 8 ? qt + 1 : 7, sn, dpn);        // (the last iteration recomputes tile 7: branch-free)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = 32 * qt + (r & 3) + 8 * (r >> 2) + 4 * hf;
                const float p = fast_exp2(fmaf(s[r], scale * 1.4426950408889634f, -lse_s[qq] * 1.4426950408889634f));
                s[r] = p;
                dp[r] = p * (dp[r] - del_s[qq]) * scale;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf, dsf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    pf[e] = (short)f2bf(s[8 * s2 + e]);
                    dsf[e] = (short)f2bf(dp[8 * s2 + e]);
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mfma_32x32x16_bf16(attb_tr_frag(dot_img, 32 * dt + lq, 2 * qt + s2, hf), pf, dv[dt]);
                    dk[dt] = mfma_32x32x16_bf16(attb_tr_frag(qt_img, 32 * dt + lq, 2 * qt + s2, hf), dsf, dk[dt]);
                }
            }
            s = sn;
            dp = dpn;
        }
        bf16_t* drow = dqkv + ((long)view * ATT_T + key) * rs3 + head * ATT_D;
        attb_store_t(drow + E, dk, hf);
        attb_store_t(drow + 2 * E, dv, hf);
        __syncthreads();                                     // the images are rewritten by the next block
    }
}

// ---- dK / dV on a double-buffered LDS-DMA image with transposing LDS reads (round 2).  The kernel above spends two thirds of a
// block outside its MFMA loop (phase profile of wave 0: issuing + waiting for the block's global loads 20 %, writing the four
// LDS images 9 %, the two block barriers 28 %, stores 8 %).  Here
//   * the transposed images are gone: the dV / dK products take their dO^T / Q^T fragments out of the ROW images with
//     ds_read_b64_tr_b16 - the MFMA's contraction order (e & 3) + 8 (e >> 2) + 4 hf over a 16-query step is exactly two such
//     reads of 4 query rows each (rows 4 hf .. + 3 and 8 + 4 hf .. + 3),
//   * the two remaining images (Q, dO rows: 64 KiB + lse / delta) fit twice: block b + 1's are written by LDS-DMA (no VGPRs, no
//     ds_write pass) into the other buffer while block b is multiplied; k / v fragments of block b + 1 wait in registers,
//   * one barrier per block (behind the drain of the DMA), no barrier after the stores.
// Row image swizzle: slot ^ f(row), f = x ^ ((x & 1) << 2), x = (row >> 1) & 7 - a bijection of the forward kernel's x (the
// ds_read_b128 row fragments stay conflict-free) whose bit 2 alternates every two rows, which puts the 4 rows x 64 B of a
// transposing half-wave read on four different 64-byte bank groups (brute-force checked: 1-way for both read kinds).
constexpr int ATTB_TR_BUF = 2 * ATTB_IMG + 2 * ATT_T * 4;      // Q rows, dO rows, lse, delta
constexpr int ATTB_DKV_TR_SMEM = 2 * ATTB_TR_BUF;              // 132 KiB
__device__ __forceinline__ int attb_swz2(int row) {
    const int x = (row >> 1) & 7;
    return x ^ ((x & 1) << 2);
}
__global__ __launch_bounds__(512) void attention_bwd_dkv_tr_kernel(const bf16_t* __restrict__ qkv,
                                                                   const bf16_t* __restrict__ d_o,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv, int heads, float scale, int nblocks, int lab) {
    char* smem = dynamic_smem();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int E = heads * ATT_D;
    const long rs3 = 3L * E;
    const int key = 32 * w + lq;
    // DMA: a 1-KiB piece = 8 image rows; lane L writes row 8 n + L / 8, slot L % 8 and fetches the slot the swizzle puts there.
    // Wave w moves pieces 4 w .. 4 w + 3 of both images and 64 of the 512 statistics.
    // (the per-lane source offsets are re-derived per block from an opaque lane id: kept in registers across the block they are
    // spilled, and every reload of a spilled register is a scratch load behind `s_waitcnt vmcnt(0)` in front of the requests)
    auto dma_block = [&](int blk, int buf) {
        const int view = blk / heads, head = blk % heads;
        const bf16_t* q_base = qkv + (long)view * ATT_T * rs3 + head * ATT_D;
        const bf16_t* do_base = d_o + (long)view * ATT_T * E + head * ATT_D;
        char* base = smem + buf * ATTB_TR_BUF;
        const int ln = opaque_vgpr((int)threadIdx.x) & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (4 * w + i) + (ln >> 3), src = (ln & 7) ^ attb_swz2(row);
            glds16(q_base + (unsigned)(row * (int)rs3 + src * 8), base + (4 * w + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (4 * w + i) + (ln >> 3), src = (ln & 7) ^ attb_swz2(row);
            glds16(do_base + (unsigned)(row * E + src * 8), base + ATTB_IMG + (4 * w + i) * 1024);
        }
        const float* stat = (w < 4 ? lse : delta) + ((long)view * heads + head) * ATT_T + 64 * (w & 3) + ln;
        glds4(stat, base + 2 * ATTB_IMG + 256 * w);          // lse_s = floats 0-255, del_s = floats 256-511
    };
    u32x4 kw[4], vw[4];
    auto request_kv = [&](int blk) {
        const int view = blk / heads, head = blk % heads;
        const bf16_t* krow = qkv + ((long)view * ATT_T + key) * rs3 + head * ATT_D + E;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kw[kk] = *reinterpret_cast<const u32x4*>(krow + 16 * kk + 8 * hf);
            vw[kk] = *reinterpret_cast<const u32x4*>(krow + E + 16 * kk + 8 * hf);
        }
    };
    // transposing reads: lane (j = lane & 15: row j / 4, 4 columns at 4 (j % 4); lanes 16-31: columns + 16; hf: rows + 4)
    unsigned troff[2][2];                                     // [d tile][rows 0-3 / 8-11 of the 16-query step]
    {
        const int j = lane & 15, g16 = (lane >> 4) & 1, r4 = j >> 2, c = j & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rsel = 0; rsel < 2; ++rsel) {
                const int row = 8 * rsel + 4 * hf + r4, d0 = 32 * dt + 16 * g16 + 4 * c;
                troff[dt][rsel] = (unsigned)(row * 128 + (((d0 >> 3) ^ attb_swz2(row)) << 4) + (d0 & 7) * 2);
            }
    }
    const unsigned smem_addr = lds_addr_of(smem);

    int blk = blockIdx.x;
    if (blk >= nblocks) return;
    dma_block(blk, 0);
    request_kv(blk);
    glds_wait_all();
    for (int it = 0; blk < nblocks; blk += gridDim.x, ++it) {
        const int view = blk / heads, head = blk % heads, buf = it & 1;
        const char* q_img = smem + buf * ATTB_TR_BUF;
        const char* do_img = q_img + ATTB_IMG;
        const float* lse_s = reinterpret_cast<const float*>(q_img + 2 * ATTB_IMG);
        const float* del_s = lse_s + ATT_T;
        // Every wave drained its requests for this block (DMA pieces, k / v rows) BEFORE it stored the previous block's results
        // (below): the barrier publishes the images and says that every wave has left the previous block, whose buffer the next
        // block's DMA overwrites - and the stores drain under this block's products instead of in front of a `vmcnt(0)`.
        // (A full drain, not a counted wait: counted waits must not span LDS-DMA and register loads, see gemm_tn384.h.)
        __syncthreads();
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kf[kk] = __builtin_bit_cast(bf16x8, kw[kk]);
            vf[kk] = __builtin_bit_cast(bf16x8, vw[kk]);
        }
        if (blk + (int)gridDim.x < nblocks && !(lab & 2)) {     // (lab bit 2: no loads after the first block)
            dma_block(blk + gridDim.x, buf ^ 1);
            request_kv(blk + gridDim.x);
        }

        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
        auto scores = [&](int qt, f32x16& s, f32x16& dp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            const int row = 32 * qt + lq;
            const int f = attb_swz2(row);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int off = row * 128 + (((2 * kk + hf) ^ f) << 4);
                s = mfma_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(q_img + off), kf[kk], s);       // S[q][key]
                dp = mfma_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(do_img + off), vf[kk], dp);    // dP[q][key]
            }
        };
        f32x16 s, dp;
        scores(0, s, dp);
        const unsigned q_addr = smem_addr + (unsigned)(buf * ATTB_TR_BUF), do_addr = q_addr + ATTB_IMG;
        // dO^T / Q^T fragments of a 16-query step, both d tiles: 8 transposing reads (prelude: lds_read_tr), retired in issue
        // order.  Step 0's are requested in front of the bf16 packing of the tile's P / dS, step 1's behind step 0's products.
#define ATTB_TR_READS(x, OFF)                                                                                         \
        lds_read_tr<OFF>(x[0][0], do_addr + tile + troff[0][0]); lds_read_tr<OFF>(x[0][1], do_addr + tile + troff[0][1]);   \
        lds_read_tr<OFF>(x[1][0], q_addr + tile + troff[0][0]);  lds_read_tr<OFF>(x[1][1], q_addr + tile + troff[0][1]);    \
        lds_read_tr<OFF>(x[2][0], do_addr + tile + troff[1][0]); lds_read_tr<OFF>(x[2][1], do_addr + tile + troff[1][1]);   \
        lds_read_tr<OFF>(x[3][0], q_addr + tile + troff[1][0]);  lds_read_tr<OFF>(x[3][1], q_addr + tile + troff[1][1]);
#pragma unroll 1
        for (int qt = 0; qt < 8; ++qt) {
            f32x16 sn, dpn;
            scores(qt + 1 < 8 ? qt + 1 : 7, sn, dpn);        // (the last iteration recomputes tile 7: branch-free)
            const unsigned tile = (unsigned)(qt * 32 * 128);
            tr_u32x2 x0[4][2], x1[4][2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = 32 * qt + (r & 3) + 8 * (r >> 2) + 4 * hf;
                const float p = fast_exp2(fmaf(s[r], scale * 1.4426950408889634f, -lse_s[qq] * 1.4426950408889634f));
                s[r] = p;
                dp[r] = p * (dp[r] - del_s[qq]) * scale;
            }
            CCD_SCHED_FENCE();
            ATTB_TR_READS(x0, 0)                             // (land under the packing below)
            bf16x8 pf, dsf;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pf[e] = (short)f2bf(s[e]);
                dsf[e] = (short)f2bf(dp[e]);
            }
            {
                bf16x8 a = frag_from_tr(x0[0][0], x0[0][1]), b = frag_from_tr(x0[1][0], x0[1][1]);
                bf16x8 c = frag_from_tr(x0[2][0], x0[2][1]), d = frag_from_tr(x0[3][0], x0[3][1]);
                // (every fragment goes through a wait of its own: a product whose operands do not depend on the wait would be
                // free to move above it)
                lds_wait_frag<6>(a);
                dv[0] = mfma_32x32x16_bf16(a, pf, dv[0]);
                lds_wait_frag<4>(b);
                dk[0] = mfma_32x32x16_bf16(b, dsf, dk[0]);
                lds_wait_frag<2>(c);
                dv[1] = mfma_32x32x16_bf16(c, pf, dv[1]);
                lds_wait_frag<0>(d);
                dk[1] = mfma_32x32x16_bf16(d, dsf, dk[1]);
            }
            CCD_SCHED_FENCE();
            ATTB_TR_READS(x1, 2048)                          // (step 1's land under step 0's four MFMAs and step 1's packing)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pf[e] = (short)f2bf(s[8 + e]);
                dsf[e] = (short)f2bf(dp[8 + e]);
            }
            {
                bf16x8 a = frag_from_tr(x1[0][0], x1[0][1]), b = frag_from_tr(x1[1][0], x1[1][1]);
                bf16x8 c = frag_from_tr(x1[2][0], x1[2][1]), d = frag_from_tr(x1[3][0], x1[3][1]);
                // (every fragment goes through a wait of its own: a product whose operands do not depend on the wait would be
                // free to move above it)
                lds_wait_frag<6>(a);
                dv[0] = mfma_32x32x16_bf16(a, pf, dv[0]);
                lds_wait_frag<4>(b);
                dk[0] = mfma_32x32x16_bf16(b, dsf, dk[0]);
                lds_wait_frag<2>(c);
                dv[1] = mfma_32x32x16_bf16(c, pf, dv[1]);
                lds_wait_frag<0>(d);
                dk[1] = mfma_32x32x16_bf16(d, dsf, dk[1]);
            }
            s = sn;
            dp = dpn;
        }
#undef ATTB_TR_READS
        glds_wait_all();                                     // the next block's requests (issued a whole block ago)
        bf16_t* drow = dqkv + ((long)view * ATT_T + key) * rs3 + head * ATT_D;
        if (!(lab & 1) || blk == (int)blockIdx.x) {          // (lab bit 1: stores of the first block only)
            attb_store_t(drow + E, dk, hf);
            attb_store_t(drow + 2 * E, dv, hf);
        }
    }
}


}  // namespace ccd
