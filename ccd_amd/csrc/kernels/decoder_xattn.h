// decoder_xattn.h - the encoder-decoder attention of the NRTR decoder on the MFMA units: <= 32 queries (the 25 target
// positions, padded) against the 256 image tokens of a sample, d_k = d_v = 64 (transformer_module.py:22-32, 84-92).
// Same interface as dec_attn_fwd/bwd_kernel (decoder.h), which stays the general path (masks, short key sequences).
//
// One workgroup (4 waves) per (sample, head); wave w owns keys 64w .. 64w+63 and keeps their K (and V) rows as MFMA
// fragments in REGISTERS, loaded straight from HBM - every K/V row is read once and never staged as a row image.
// Transposed-product trick of attention_fwd.h: S^T = K.Q^T leaves a lane holding scores of ONE query, the
// exponentiated accumulators are directly the B operand of the next product, and the LDS-resident A operands
// (V^T, K^T, Q^T, dO^T images) carry the matching key / query permutation.
//   forward : 16 MFMAs per wave; softmax statistics of the 4 key slices are merged through LDS BEFORE the P.V product,
//             so the partial outputs only need a sum.
//   backward: the scores are formed in both orientations (lane = query for dQ, lane = key for dK / dV): 56 MFMAs/wave.
// HBM-bound: algorithmic bytes per (sample, head) = 2 * 256 * 64 * 2 B read (+ the same written in the backward).
#pragma once

namespace ccd {

constexpr int XA_TQ = 32, XA_TK = 256, XA_D = 64;
constexpr int XA_MERGE = 4 * XA_TQ * 68 * 4;                           // per-wave fp32 [32 q][64 d (+4 pad)]
constexpr int XA_FWD_SMEM = XA_MERGE + 2 * 4 * XA_TQ * 4;               // V^T image / merge buffer + per-wave (max, sum)
constexpr int XA_QT_IMG = XA_D * XA_TQ * 2;                            // [64 d][32 q] bf16
constexpr int XA_BWD_SMEM = XA_MERGE + 2 * XA_QT_IMG + 2 * XA_TQ * 4;   // K^T image / merge buffer, Q^T, dO^T, lse, delta
static_assert(XA_MERGE >= XA_TK * XA_D * 2, "merge buffer overlays the transposed image");

// [64 d][32 q] transposed image of a [Tq, 64] slab; q permuted inside 16-groups like the keys of att_stage_transposed,
// 16-byte slots XORed with (d >> 1) & 3 (2-way instead of 8-way bank conflicts on the fragment reads)
__device__ __forceinline__ void xa_stage_qt(const bf16_t* __restrict__ src, long ld, int Tq, char* img) {
    const int q = threadIdx.x >> 3, c = (threadIdx.x & 7) * 8;
    u32x4 w = {0u, 0u, 0u, 0u};
    if (q < Tq) w = *reinterpret_cast<const u32x4*>(src + (long)q * ld + c);
    const int pos = 16 * (q >> 4) + 4 * att_chunk_pos((q >> 2) & 3) + (q & 3);      // k-slot of query q
    const int slot = pos >> 3, within = pos & 7;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int d = c + e;
        const unsigned word = w[e >> 1];
        const bf16_t v = (bf16_t)((e & 1) ? (word >> 16) : (word & 0xffffu));
        *reinterpret_cast<bf16_t*>(img + d * 64 + ((slot ^ ((d >> 1) & 3)) * 16) + within * 2) = v;
    }
}
__device__ __forceinline__ bf16x8 xa_qt_frag(const char* img, int d, int s2, int hf) {
    return *reinterpret_cast<const bf16x8*>(img + d * 64 + (((2 * s2 + hf) ^ ((d >> 1) & 3)) * 16));
}
__device__ __forceinline__ bf16x8 xa_zero_frag() {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = 0;
    return z;
}
// key of accumulator register r in tile kt for a lane of half hf (rows of the 32x32 MFMA result)
__device__ __forceinline__ int xa_acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

__global__ __launch_bounds__(256, 4) void xattn_fwd_kernel(DecAttnParams p) {
    char* smem = dynamic_smem();
    char* vt_img = smem;                                               // later: the merge buffer
    float* st_m = reinterpret_cast<float*>(smem + XA_MERGE);
    float* st_l = st_m + 4 * XA_TQ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int Tq = p.Tq;
    const bf16_t* k_base = p.k + (long)b * XA_TK * p.ldk + h * XA_D;
    const bf16_t* v_base = p.v + (long)b * XA_TK * p.ldv + h * XA_D;
    bf16x8 kf[2][4], qf[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            kf[t][kk] = *reinterpret_cast<const bf16x8*>(k_base + (long)(64 * w + 32 * t + lq) * p.ldk + 16 * kk + 8 * hf);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        qf[kk] = lq < Tq ? *reinterpret_cast<const bf16x8*>(p.q + ((long)b * Tq + lq) * p.ldq + h * XA_D + 16 * kk + 8 * hf)
                         : xa_zero_frag();
    att_stage_transposed(v_base, p.ldv, vt_img);
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s[t] = mfma_32x32x16_bf16(kf[t][kk], qf[kk], s[t]);
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    mx = fmaxf(mx, shfl_xor(mx, 32));
    if (hf == 0) st_m[w * XA_TQ + lq] = mx;
    __syncthreads();                                                   // also: the V^T image is complete
    const float gm = fmaxf(fmaxf(st_m[lq], st_m[XA_TQ + lq]), fmaxf(st_m[2 * XA_TQ + lq], st_m[3 * XA_TQ + lq]));
    const float c2 = p.scale * 1.4426950408889634f;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = fast_exp2((s[t][r] - gm) * c2);
            s[t][r] = e;
            sum += e;
        }
    sum += shfl_xor(sum, 32);
    if (hf == 0) st_l[w * XA_TQ + lq] = sum;
    __syncthreads();
    const float total = st_l[lq] + st_l[XA_TQ + lq] + st_l[2 * XA_TQ + lq] + st_l[3 * XA_TQ + lq];
    const float inv = 1.0f / total;
    const long prow = (((long)b * p.H + h) * Tq + lq) * XA_TK;
    if (w == 0 && hf == 0 && lq < Tq) p.lse[((long)b * p.H + h) * Tq + lq] = gm * p.scale + logf(total);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pr = s[t][r] * inv;
            const int key = 64 * w + 32 * t + xa_acc_row(r, hf);
            if (p.thr) pr = drop_keep(p.seed, (unsigned long long)(prow + key), p.thr) ? pr * p.keep_scale : 0.f;
            s[t][r] = pr;
            if (p.probs && lq < Tq) p.probs[prow + key] = pr;
        }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (short)f2bf(s[t][8 * s2 + e]);
            const int ks = 2 * (2 * w + t) + s2;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int d = 32 * dt + lq;
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vt_img + d * 512 + (((2 * ks + hf) ^ (d & 15)) * 16));
                o[dt] = mfma_32x32x16_bf16(vf, pf, o[dt]);
            }
        }
    __syncthreads();                                                   // every wave is done with the V^T image
    float* mg = reinterpret_cast<float*>(smem) + w * XA_TQ * 68;        // this wave's [32 q][68] partial output
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4v v4;
            v4.x = o[dt][4 * g]; v4.y = o[dt][4 * g + 1]; v4.z = o[dt][4 * g + 2]; v4.w = o[dt][4 * g + 3];
            *reinterpret_cast<f32x4v*>(mg + lq * 68 + 32 * dt + 8 * g + 4 * hf) = v4;
        }
    __syncthreads();
    {
        const int q = threadIdx.x >> 3, c = (threadIdx.x & 7) * 8;
        if (q < Tq) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            const float* base = reinterpret_cast<const float*>(smem) + q * 68 + c;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const f32x4v a = *reinterpret_cast<const f32x4v*>(base + ww * XA_TQ * 68);
                const f32x4v bq = *reinterpret_cast<const f32x4v*>(base + ww * XA_TQ * 68 + 4);
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                acc[4] += bq.x; acc[5] += bq.y; acc[6] += bq.z; acc[7] += bq.w;
            }
            *reinterpret_cast<u32x4*>(p.out + ((long)b * Tq + q) * p.ldo + h * XA_D + c) = pack8(acc);
        }
    }
}

__global__ __launch_bounds__(256, 2) void xattn_bwd_kernel(DecAttnParams p) {
    char* smem = dynamic_smem();
    char* kt_img = smem;                                               // later: the dQ merge buffer
    char* qt_img = smem + XA_MERGE;
    char* dot_img = qt_img + XA_QT_IMG;
    float* lse_s = reinterpret_cast<float*>(dot_img + XA_QT_IMG);
    float* del_s = lse_s + XA_TQ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int Tq = p.Tq;
    const bf16_t* k_base = p.k + (long)b * XA_TK * p.ldk + h * XA_D;
    const bf16_t* v_base = p.v + (long)b * XA_TK * p.ldv + h * XA_D;
    const bf16_t* q_base = p.q + (long)b * Tq * p.ldq + h * XA_D;
    const bf16_t* do_base = p.d_out + (long)b * Tq * p.ldo + h * XA_D;
    bf16x8 kf[2][4], vf[2][4], qf[4], dof[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const long row = 64 * w + 32 * t + lq;
            kf[t][kk] = *reinterpret_cast<const bf16x8*>(k_base + row * p.ldk + 16 * kk + 8 * hf);
            vf[t][kk] = *reinterpret_cast<const bf16x8*>(v_base + row * p.ldv + 16 * kk + 8 * hf);
        }
    float dsum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (lq < Tq) {
            qf[kk] = *reinterpret_cast<const bf16x8*>(q_base + (long)lq * p.ldq + 16 * kk + 8 * hf);
            const u32x4 dw = *reinterpret_cast<const u32x4*>(do_base + (long)lq * p.ldo + 16 * kk + 8 * hf);
            const u32x4 ow = *reinterpret_cast<const u32x4*>(p.out + ((long)b * Tq + lq) * p.ldo + h * XA_D + 16 * kk + 8 * hf);
            dof[kk] = __builtin_bit_cast(bf16x8, dw);
            float a[8], c[8];
            unpack8(dw, a);
            unpack8(ow, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += a[e] * c[e];
        } else {
            qf[kk] = xa_zero_frag();
            dof[kk] = xa_zero_frag();
        }
    }
    dsum += shfl_xor(dsum, 32);                                        // delta[q] = <d_out[q], out[q]>
    const float my_lse = lq < Tq ? p.lse[((long)b * p.H + h) * Tq + lq] : 0.f;
    if (w == 0 && hf == 0) { lse_s[lq] = my_lse; del_s[lq] = dsum; }
    att_stage_transposed(k_base, p.ldk, kt_img);
    xa_stage_qt(q_base, p.ldq, Tq, qt_img);
    xa_stage_qt(do_base, p.ldo, Tq, dot_img);
    __syncthreads();
    const float c2 = p.scale * 1.4426950408889634f, l2e = 1.4426950408889634f;
    const long pbase = ((long)b * p.H + h) * Tq;
    // ---- orientation 1: lane = query.  dS^T tiles feed dQ^T[d][q] += K^T[d][keys] . dS^T[keys][q]
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = mfma_32x32x16_bf16(kf[t][kk], qf[kk], s);
            dp = mfma_32x32x16_bf16(vf[t][kk], dof[kk], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = fast_exp2(fmaf(s[r], c2, -my_lse * l2e));
            float g = dp[r];
            if (p.thr) {
                const int key = 64 * w + 32 * t + xa_acc_row(r, hf);
                g = drop_keep(p.seed, (unsigned long long)((pbase + lq) * XA_TK + key), p.thr) ? g * p.keep_scale : 0.f;
            }
            s[r] = pr * (g - dsum) * p.scale;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 dsf;
#pragma unroll
            for (int e = 0; e < 8; ++e) dsf[e] = (short)f2bf(s[8 * s2 + e]);
            const int ks = 2 * (2 * w + t) + s2;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int d = 32 * dt + lq;
                const bf16x8 ktf = *reinterpret_cast<const bf16x8*>(kt_img + d * 512 + (((2 * ks + hf) ^ (d & 15)) * 16));
                dq[dt] = mfma_32x32x16_bf16(ktf, dsf, dq[dt]);
            }
        }
    }
    // ---- orientation 2: lane = key.  P[q][key], dS[q][key] feed dV^T[d][key] += dO^T[d][q] . Pd, dK^T += Q^T . dS
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = mfma_32x32x16_bf16(qf[kk], kf[t][kk], s);
            dp = mfma_32x32x16_bf16(dof[kk], vf[t][kk], dp);
        }
        const int key = 64 * w + 32 * t + lq;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = xa_acc_row(r, hf);
            float pr = qq < Tq ? fast_exp2(fmaf(s[r], c2, -lse_s[qq] * l2e)) : 0.f;
            float g = dp[r], pd = pr;
            if (p.thr) {
                const bool keep = drop_keep(p.seed, (unsigned long long)((pbase + qq) * XA_TK + key), p.thr);
                g = keep ? g * p.keep_scale : 0.f;
                pd = keep ? pr * p.keep_scale : 0.f;
            }
            s[r] = pd;
            dp[r] = pr * (g - del_s[qq]) * p.scale;
        }
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
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
                dv[dt] = mfma_32x32x16_bf16(xa_qt_frag(dot_img, 32 * dt + lq, s2, hf), pf, dv[dt]);
                dk[dt] = mfma_32x32x16_bf16(xa_qt_frag(qt_img, 32 * dt + lq, s2, hf), dsf, dk[dt]);
            }
        }
        attb_store_t(p.dk + ((long)b * XA_TK + key) * p.lddk + h * XA_D, dk, hf);
        attb_store_t(p.dv + ((long)b * XA_TK + key) * p.lddv + h * XA_D, dv, hf);
    }
    __syncthreads();                                                   // every wave is done with the K^T image
    float* mg = reinterpret_cast<float*>(smem) + w * XA_TQ * 68;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4v v4;
            v4.x = dq[dt][4 * g]; v4.y = dq[dt][4 * g + 1]; v4.z = dq[dt][4 * g + 2]; v4.w = dq[dt][4 * g + 3];
            *reinterpret_cast<f32x4v*>(mg + lq * 68 + 32 * dt + 8 * g + 4 * hf) = v4;
        }
    __syncthreads();
    {
        const int q = threadIdx.x >> 3, c = (threadIdx.x & 7) * 8;
        if (q < Tq) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            const float* base = reinterpret_cast<const float*>(smem) + q * 68 + c;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const f32x4v a = *reinterpret_cast<const f32x4v*>(base + ww * XA_TQ * 68);
                const f32x4v bq = *reinterpret_cast<const f32x4v*>(base + ww * XA_TQ * 68 + 4);
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                acc[4] += bq.x; acc[5] += bq.y; acc[6] += bq.z; acc[7] += bq.w;
            }
            *reinterpret_cast<u32x4*>(p.dq + ((long)b * Tq + q) * p.lddq + h * XA_D + c) = pack8(acc);
        }
    }
}

// ------------------------------------------------------------------------------------- masked self-attention
// <= 32 queries x <= 32 keys (the target sequence against itself, pad & causal mask, nrtr_decoder.py:77-104): ONE
// 32x32 score tile per (sample, head), so one WAVE owns a (sample, head) and a workgroup handles four of them; no
// cross-wave traffic at all.  Same operand conventions as above.
constexpr int SA_WAVE_FWD = XA_QT_IMG;                                  // V^T image
constexpr int SA_WAVE_BWD = 3 * XA_QT_IMG + 2 * XA_TQ * 4;              // K^T, Q^T, dO^T images, lse, delta
constexpr int SA_FWD_SMEM = 4 * SA_WAVE_FWD, SA_BWD_SMEM = 4 * SA_WAVE_BWD;

__device__ __forceinline__ void sa_stage_t(const bf16_t* __restrict__ src, long ld, int T, char* img, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = lane + 64 * i, q = id >> 3, c = (id & 7) * 8;
        u32x4 w = {0u, 0u, 0u, 0u};
        if (q < T) w = *reinterpret_cast<const u32x4*>(src + (long)q * ld + c);
        const int pos = 16 * (q >> 4) + 4 * att_chunk_pos((q >> 2) & 3) + (q & 3);
        const int slot = pos >> 3, within = pos & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = c + e;
            const unsigned word = w[e >> 1];
            const bf16_t v = (bf16_t)((e & 1) ? (word >> 16) : (word & 0xffffu));
            *reinterpret_cast<bf16_t*>(img + d * 64 + ((slot ^ ((d >> 1) & 3)) * 16) + within * 2) = v;
        }
    }
}
// bit j set <=> key j may be attended to at all (inside the sequence, not <PAD>, inside key_len)
__device__ __forceinline__ unsigned sa_key_mask(const DecAttnParams& p, int b, int lane) {
    bool vis = lane < p.Tk;
    if (vis && p.key_len) vis = lane < p.key_len[b];
    if (vis && p.tokens) vis = p.tokens[(long)b * p.Tk + lane] != (long long)p.pad_idx;
    return (unsigned)(ballot(vis) & 0xffffffffull);
}
__device__ __forceinline__ void sa_load_rows(const bf16_t* __restrict__ base, long ld, int T, int lq, int hf, bf16x8 (&f)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        f[kk] = lq < T ? *reinterpret_cast<const bf16x8*>(base + (long)lq * ld + 16 * kk + 8 * hf) : xa_zero_frag();
}

__global__ __launch_bounds__(256) void sattn_fwd_kernel(DecAttnParams p) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int item = blockIdx.x * 4 + w;
    if (item >= p.B * p.H) return;                                     // wave-uniform; no workgroup barriers below
    char* vt_img = dynamic_smem() + w * SA_WAVE_FWD;
    const int b = item / p.H, h = item % p.H;
    const int Tq = p.Tq, Tk = p.Tk;
    bf16x8 kf[4], qf[4];
    sa_load_rows(p.k + (long)b * Tk * p.ldk + h * XA_D, p.ldk, Tk, lq, hf, kf);
    sa_load_rows(p.q + (long)b * Tq * p.ldq + h * XA_D, p.ldq, Tq, lq, hf, qf);
    sa_stage_t(p.v + (long)b * Tk * p.ldv + h * XA_D, p.ldv, Tk, vt_img, lane);
    const unsigned kmask = sa_key_mask(p, b, lane);
    wave_lds_fence();                                                  // the wave's own V^T image is complete
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) s = mfma_32x32x16_bf16(kf[kk], qf[kk], s);
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = xa_acc_row(r, hf);
        const bool vis = ((kmask >> key) & 1u) && (!p.causal || key <= lq);
        s[r] = vis ? s[r] : -3.0e38f;
        mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, shfl_xor(mx, 32));
    const bool dead = mx < -1.0e37f;                                   // nothing visible (reference: NaN row)
    const float c2 = p.scale * 1.4426950408889634f;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e = (dead || s[r] < -1.0e37f) ? 0.f : fast_exp2((s[r] - mx) * c2);
        s[r] = e;
        sum += e;
    }
    sum += shfl_xor(sum, 32);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    const long prow = (((long)b * p.H + h) * Tq + lq) * Tk;
    if (hf == 0 && lq < Tq) p.lse[((long)b * p.H + h) * Tq + lq] = sum > 0.f ? mx * p.scale + logf(sum) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float pr = s[r] * inv;
        const int key = xa_acc_row(r, hf);
        if (p.thr) pr = drop_keep(p.seed, (unsigned long long)(prow + key), p.thr) ? pr * p.keep_scale : 0.f;
        s[r] = pr;
        if (p.probs && lq < Tq && key < Tk) p.probs[prow + key] = pr;
    }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (short)f2bf(s[8 * s2 + e]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = mfma_32x32x16_bf16(xa_qt_frag(vt_img, 32 * dt + lq, s2, hf), pf, o[dt]);
    }
    attb_store_t(p.out + ((long)b * Tq + lq) * p.ldo + h * XA_D, o, hf, lq < Tq);     // (every lane takes part in the exchange)
}

__global__ __launch_bounds__(256) void sattn_bwd_kernel(DecAttnParams p) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int item = blockIdx.x * 4 + w;
    if (item >= p.B * p.H) return;
    char* kt_img = dynamic_smem() + w * SA_WAVE_BWD;
    char* qt_img = kt_img + XA_QT_IMG;
    char* dot_img = qt_img + XA_QT_IMG;
    float* lse_s = reinterpret_cast<float*>(dot_img + XA_QT_IMG);
    float* del_s = lse_s + XA_TQ;
    const int b = item / p.H, h = item % p.H;
    const int Tq = p.Tq, Tk = p.Tk;
    const bf16_t* k_base = p.k + (long)b * Tk * p.ldk + h * XA_D;
    const bf16_t* q_base = p.q + (long)b * Tq * p.ldq + h * XA_D;
    const bf16_t* do_base = p.d_out + (long)b * Tq * p.ldo + h * XA_D;
    bf16x8 kf[4], vf[4], qf[4], dof[4];
    sa_load_rows(k_base, p.ldk, Tk, lq, hf, kf);
    sa_load_rows(p.v + (long)b * Tk * p.ldv + h * XA_D, p.ldv, Tk, lq, hf, vf);
    sa_load_rows(q_base, p.ldq, Tq, lq, hf, qf);
    sa_load_rows(do_base, p.ldo, Tq, lq, hf, dof);
    float dsum = 0.f;
    if (lq < Tq) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const u32x4 ow = *reinterpret_cast<const u32x4*>(p.out + ((long)b * Tq + lq) * p.ldo + h * XA_D + 16 * kk + 8 * hf);
            float a[8], c[8];
            unpack8(__builtin_bit_cast(u32x4, dof[kk]), a);
            unpack8(ow, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += a[e] * c[e];
        }
    }
    dsum += shfl_xor(dsum, 32);
    const float my_lse = lq < Tq ? p.lse[((long)b * p.H + h) * Tq + lq] : 0.f;
    if (hf == 0) { lse_s[lq] = my_lse; del_s[lq] = dsum; }
    sa_stage_t(k_base, p.ldk, Tk, kt_img, lane);
    sa_stage_t(q_base, p.ldq, Tq, qt_img, lane);
    sa_stage_t(do_base, p.ldo, Tq, dot_img, lane);
    const unsigned kmask = sa_key_mask(p, b, lane);
    wave_lds_fence();                                                  // the wave's own images / statistics are complete
    const float c2 = p.scale * 1.4426950408889634f, l2e = 1.4426950408889634f;
    const long pbase = ((long)b * p.H + h) * Tq;
    {   // orientation 1: lane = query -> dQ
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = mfma_32x32x16_bf16(kf[kk], qf[kk], s);
            dp = mfma_32x32x16_bf16(vf[kk], dof[kk], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = xa_acc_row(r, hf);
            const bool vis = lq < Tq && ((kmask >> key) & 1u) && (!p.causal || key <= lq);
            const float pr = vis ? fast_exp2(fmaf(s[r], c2, -my_lse * l2e)) : 0.f;
            float g = dp[r];
            if (p.thr) g = drop_keep(p.seed, (unsigned long long)((pbase + lq) * Tk + key), p.thr) ? g * p.keep_scale : 0.f;
            s[r] = pr * (g - dsum) * p.scale;
        }
        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 dsf;
#pragma unroll
            for (int e = 0; e < 8; ++e) dsf[e] = (short)f2bf(s[8 * s2 + e]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma_32x32x16_bf16(xa_qt_frag(kt_img, 32 * dt + lq, s2, hf), dsf, dq[dt]);
        }
        attb_store_t(p.dq + ((long)b * Tq + lq) * p.lddq + h * XA_D, dq, hf, lq < Tq);
    }
    {   // orientation 2: lane = key -> dK, dV
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = mfma_32x32x16_bf16(qf[kk], kf[kk], s);
            dp = mfma_32x32x16_bf16(dof[kk], vf[kk], dp);
        }
        const bool key_vis = (kmask >> lq) & 1u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = xa_acc_row(r, hf);
            const bool vis = qq < Tq && key_vis && (!p.causal || lq <= qq);
            const float pr = vis ? fast_exp2(fmaf(s[r], c2, -lse_s[qq] * l2e)) : 0.f;
            float g = dp[r], pd = pr;
            if (p.thr) {
                const bool keep = drop_keep(p.seed, (unsigned long long)((pbase + qq) * Tk + lq), p.thr);
                g = keep ? g * p.keep_scale : 0.f;
                pd = keep ? pr * p.keep_scale : 0.f;
            }
            s[r] = pd;
            dp[r] = pr * (g - del_s[qq]) * p.scale;
        }
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
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
                dv[dt] = mfma_32x32x16_bf16(xa_qt_frag(dot_img, 32 * dt + lq, s2, hf), pf, dv[dt]);
                dk[dt] = mfma_32x32x16_bf16(xa_qt_frag(qt_img, 32 * dt + lq, s2, hf), dsf, dk[dt]);
            }
        }
        attb_store_t(p.dk + ((long)b * Tk + lq) * p.lddk + h * XA_D, dk, hf, lq < Tk);
        attb_store_t(p.dv + ((long)b * Tk + lq) * p.lddv + h * XA_D, dv, hf, lq < Tk);
    }
}

}  // namespace ccd
