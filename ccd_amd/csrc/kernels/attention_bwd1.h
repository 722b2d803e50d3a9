// attention_bwd1.h - round 4: the backward of the T = 256, d = 64 attention in ONE pass per (view, head): five products instead of the
// seven of attention_bwd.h's kernel pair (which forms S, dP and the softmax twice - once per orientation - and reads q, k, v, dO twice:
// 1.23 GB of HBM traffic per layer for 0.81 GB of operands).
//   S = Q K^T, dP = dO V^T, P = exp(S * scale - lse), dS = P * (dP - delta) * scale;   dV = P^T dO, dK = dS^T Q, dQ = dS K
// (vision_transformer.py:80-92 run backwards by autograd).
// A workgroup = 8 waves = one (view, head); wave w owns KEYS 32 w .. + 31 exactly as attention_bwd_dkv_tr_kernel does: S and dP come
// out of the MFMA as [query rows][key column = lane], P and dS are the B operands of the dV / dK products straight from the registers.
// That orientation cannot feed dQ (its contraction runs over keys = lanes), so each wave also WRITES its bf16 dS tile into an LDS
// exchange image [key][query] (2 KiB per wave and 32-query step, double-buffered), and behind ONE barrier per step all eight waves
// read the eight tiles back as the k-major operand of a 16x16x32 product: wave (d16 = w & 3, q16 = w >> 2) forms the complete
// [16 d][16 queries] piece of dQ over all 256 keys (8 MFMAs, the K^T fragments of its 16 d columns stay in registers for the whole
// block) and stores it - 8 bytes per lane, the four d16 waves of a query row fill its 128-byte line together.
// delta = rowsum(dO * O) is formed here as well (the old dQ kernel's job).  LDS: Q, dO, K row images (32 KiB each, LDS-DMA, the
// swizzle of attention_bwd_dkv_tr_kernel), the exchange image (32 KiB; holds the V rows while the block starts), statistics.
#pragma once

namespace ccd {

constexpr int ATTB1_XBUF = 16384;                                        // one exchange buffer: 8 waves x [32 keys][64 B]
constexpr int ATTB1_SMEM = 3 * ATTB_IMG + 2 * ATTB1_XBUF + 2 * ATT_T * 4 + ATTB_CS_BYTES;
// 8-byte slot swizzle of an exchange row (32 queries = 8 slots): writes (16 keys x one slot) and transposing reads (4 keys x 4 slots
// per 16-lane group, two groups per LDS cycle) both touch every bank once
__device__ __forceinline__ int attb1_fx(int key) { return ((key >> 2) & 3) | ((key >> 1) & 4); }

__global__ __launch_bounds__(512) void attention_bwd_onepass_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                    const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                    bf16_t* __restrict__ dqkv, float* __restrict__ bias_ws,
                                                                    int heads, float scale, int nblocks) {
    char* smem = dynamic_smem();
    char* q_img = smem;
    char* do_img = smem + ATTB_IMG;
    char* k_img = smem + 2 * ATTB_IMG;
    char* x_buf = smem + 3 * ATTB_IMG;                                   // [2][8 waves][32 keys][64 B]; V rows at block start
    float* lse_s = reinterpret_cast<float*>(smem + 3 * ATTB_IMG + 2 * ATTB1_XBUF);
    float* del_s = lse_s + ATT_T;
    float* cs = del_s + ATT_T;                                           // [heads][64] column sums of dQ over this workgroup's blocks
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int E = heads * ATT_D;
    const long rs3 = 3L * E;
    const int key = 32 * w + lq;
    const int d16 = w & 3, q16 = w >> 2, g4 = lane >> 4, l16 = lane & 15;
    if (bias_ws)
        for (int i = threadIdx.x; i < E; i += ATTB_THREADS) cs[i] = 0.f;    // (published by the first block's barrier)

    // LDS-DMA of a block's four row images (1-KiB piece = 8 rows; wave w moves pieces 4 w .. 4 w + 3 of each) and of lse
    auto dma_block = [&](int blk) {
        const int view = blk / heads, head = blk % heads;
        const bf16_t* q_base = qkv + (long)view * ATT_T * rs3 + head * ATT_D;
        const bf16_t* do_base = d_o + (long)view * ATT_T * E + head * ATT_D;
        const int ln = opaque_vgpr((int)threadIdx.x) & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (4 * w + i) + (ln >> 3), src = (ln & 7) ^ attb_swz2(row);
            glds16(q_base + (unsigned)(row * (int)rs3 + src * 8), q_img + (4 * w + i) * 1024);
            glds16(do_base + (unsigned)(row * E + src * 8), do_img + (4 * w + i) * 1024);
            glds16(q_base + E + (unsigned)(row * (int)rs3 + src * 8), k_img + (4 * w + i) * 1024);
            glds16(q_base + 2 * E + (unsigned)(row * (int)rs3 + src * 8), x_buf + (4 * w + i) * 1024);
        }
        if (w < 4) glds4(lse + ((long)view * heads + head) * ATT_T + 64 * w + ln, reinterpret_cast<char*>(lse_s) + 256 * w);
    };
    // transposing reads of the dV / dK products (attention_bwd_dkv_tr_kernel's): dO^T / Q^T fragments out of the row images
    unsigned troff[2][2];                                                // [d tile][rows 0-3 / 8-11 of the 16-query step]
    {
        const int r4 = l16 >> 2, c = l16 & 3, g16 = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rsel = 0; rsel < 2; ++rsel) {
                const int row = 8 * rsel + 4 * hf + r4, d0 = 32 * dt + 16 * g16 + 4 * c;
                troff[dt][rsel] = (unsigned)(row * 128 + (((d0 >> 3) ^ attb_swz2(row)) << 4) + (d0 & 7) * 2);
            }
    }
    // dQ's operands (16x16x32: A[i = l16][k = 8 g4 + e], B[k][j = l16]): a 16-lane group reads 4 keys x 16 columns per transposing read
    //   K^T fragment of key tile ks: keys 32 ks + 8 g4 + 4 rsel + (l16 >> 2), d columns 16 d16 + 4 (l16 & 3) .. of the K row image
    //   dS^T fragment of key tile ks: the same keys, query columns 16 q16 + 4 (l16 & 3) .. of wave ks's exchange tile
    unsigned xoff[2];
#pragma unroll
    for (int rsel = 0; rsel < 2; ++rsel) {
        const int kk = 8 * g4 + 4 * rsel + (l16 >> 2);
        xoff[rsel] = (unsigned)(kk * 64 + (((4 * q16 + (l16 & 3)) ^ attb1_fx(kk)) << 3));
    }
    const unsigned smem_addr = lds_addr_of(smem);
    const unsigned q_addr = smem_addr, do_addr = smem_addr + ATTB_IMG, k_addr = smem_addr + 2 * ATTB_IMG, x_addr = smem_addr + 3 * ATTB_IMG;
    float csq[4] = {0.f, 0.f, 0.f, 0.f};
    int cs_head = -1;
    auto flush_cs = [&]() {                                              // this wave's dQ column sums of the blocks of one head
        if (bias_ws && cs_head >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = csq[r];
                v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4); v += shfl_xor(v, 8);
                if (l16 == 0) atomicAdd(cs + cs_head * ATT_D + 16 * d16 + 4 * g4 + r, v);
                csq[r] = 0.f;
            }
        }
    };

    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int view = blk / heads, head = blk % heads;
        if (head != cs_head) { flush_cs(); cs_head = head; }
        dma_block(blk);
        u32x4 ow[4];
        {
            const long orow = ((long)view * ATT_T + key) * E + head * ATT_D;     // (query index = this thread's key index: 32 w + lq)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) ow[kk] = *reinterpret_cast<const u32x4*>(o + orow + 16 * kk + 8 * hf);
        }
        glds_wait_all();
        lds_barrier();
        // ---- per block: k / v fragments of this wave's keys, delta of its 32 queries, the K^T fragments of its 16 d columns
        bf16x8 kf[4], vf[4], kT[8];
        {
            const int f = attb_swz2(key);
            float dsum = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int off = key * 128 + (((2 * kk + hf) ^ f) << 4);
                kf[kk] = *reinterpret_cast<const bf16x8*>(k_img + off);
                vf[kk] = *reinterpret_cast<const bf16x8*>(x_buf + off);
                const u32x4 dw = *reinterpret_cast<const u32x4*>(do_img + off);
                float a[8], b[8];
                unpack8(dw, a);
                unpack8(ow[kk], b);
#pragma unroll
                for (int e = 0; e < 8; ++e) dsum += a[e] * b[e];
            }
            dsum += shfl_xor(dsum, 32);
            if (hf == 0) del_s[key] = dsum;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                tr_u32x2 lo, hi;
                const int r4 = l16 >> 2, d0 = 16 * d16 + 4 * (l16 & 3);
                const int row0 = 32 * ks + 8 * g4 + r4, row1 = row0 + 4;
                lds_read_tr<0>(lo, k_addr + (unsigned)(row0 * 128 + (((d0 >> 3) ^ attb_swz2(row0)) << 4) + (d0 & 7) * 2));
                lds_read_tr<0>(hi, k_addr + (unsigned)(row1 * 128 + (((d0 >> 3) ^ attb_swz2(row1)) << 4) + (d0 & 7) * 2));
                lds_drain();
                kT[ks] = frag_from_tr(lo, hi);
            }
        }
        lds_barrier();                                       // delta is published; the V rows in the exchange image have been read

        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
#pragma unroll 1
        for (int qt = 0; qt < 8; ++qt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            {
                const int row = 32 * qt + lq, f = attb_swz2(row);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int off = row * 128 + (((2 * kk + hf) ^ f) << 4);
                    s = mfma_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(q_img + off), kf[kk], s);       // S[q][key]
                    dp = mfma_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(do_img + off), vf[kk], dp);    // dP[q][key]
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = 32 * qt + (r & 3) + 8 * (r >> 2) + 4 * hf;
                const float p = fast_exp2(fmaf(s[r], scale * 1.4426950408889634f, -lse_s[qq] * 1.4426950408889634f));
                s[r] = p;
                dp[r] = p * (dp[r] - del_s[qq]) * scale;
            }
            // the dS tile for the dQ product: [key = this lane][queries 8 j + 4 hf .. + 3] = 8 bytes, slot 2 j + hf of the key's row
            char* xw = x_buf + (qt & 1) * ATTB1_XBUF + w * 2048 + lq * 64;
            const int fx = attb1_fx(lq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x2 pk;
                pk.x = pack_bf2(dp[4 * j], dp[4 * j + 1]);
                pk.y = pack_bf2(dp[4 * j + 2], dp[4 * j + 3]);
                *reinterpret_cast<u32x2*>(xw + (((2 * j + hf) ^ fx) << 3)) = pk;
            }
            const unsigned tile = (unsigned)(qt * 32 * 128);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                tr_u32x2 x[4][2];
                const unsigned t2 = tile + (unsigned)(s2 * 2048);
                lds_read_tr<0>(x[0][0], do_addr + t2 + troff[0][0]); lds_read_tr<0>(x[0][1], do_addr + t2 + troff[0][1]);
                lds_read_tr<0>(x[1][0], q_addr + t2 + troff[0][0]);  lds_read_tr<0>(x[1][1], q_addr + t2 + troff[0][1]);
                lds_read_tr<0>(x[2][0], do_addr + t2 + troff[1][0]); lds_read_tr<0>(x[2][1], do_addr + t2 + troff[1][1]);
                lds_read_tr<0>(x[3][0], q_addr + t2 + troff[1][0]);  lds_read_tr<0>(x[3][1], q_addr + t2 + troff[1][1]);
                bf16x8 pf, dsf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    pf[e] = (short)f2bf(s[8 * s2 + e]);
                    dsf[e] = (short)f2bf(dp[8 * s2 + e]);
                }
                bf16x8 a = frag_from_tr(x[0][0], x[0][1]), b = frag_from_tr(x[1][0], x[1][1]);
                bf16x8 c = frag_from_tr(x[2][0], x[2][1]), d = frag_from_tr(x[3][0], x[3][1]);
                lds_wait_frag<6>(a);
                dv[0] = mfma_32x32x16_bf16(a, pf, dv[0]);
                lds_wait_frag<4>(b);
                dk[0] = mfma_32x32x16_bf16(b, dsf, dk[0]);
                lds_wait_frag<2>(c);
                dv[1] = mfma_32x32x16_bf16(c, pf, dv[1]);
                lds_wait_frag<0>(d);
                dk[1] = mfma_32x32x16_bf16(d, dsf, dk[1]);
            }
            lds_barrier();                                   // every wave's dS tile of this step is in the exchange image (LDS-only
                                                             // barrier: the dQ stores of the previous step stay in flight)
            // ---- dQ piece [16 d][16 queries] of this step over all 256 keys
            f32x4 dq = {0.f, 0.f, 0.f, 0.f};
            const unsigned xb = x_addr + (unsigned)((qt & 1) * ATTB1_XBUF);
            tr_u32x2 xl[8], xh[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                lds_read_tr<0>(xl[ks], xb + (unsigned)(ks * 2048) + xoff[0]);
                lds_read_tr<0>(xh[ks], xb + (unsigned)(ks * 2048) + xoff[1]);
            }
            mlp_static_for<0, 8>([&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                bf16x8 bfr = frag_from_tr(xl[ks], xh[ks]);
                lds_wait_frag<2 * (7 - ks)>(bfr);
                dq = mfma_16x16x32_bf16(kT[ks], bfr, dq);
            });
            {
                u32x2 pk;
                pk.x = pack_bf2(dq[0], dq[1]);
                pk.y = pack_bf2(dq[2], dq[3]);
                bf16_t* dst = dqkv + ((long)view * ATT_T + 32 * qt + 16 * q16 + l16) * rs3 + head * ATT_D + 16 * d16 + 4 * g4;
                *reinterpret_cast<u32x2*>(dst) = pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) csq[r] += dq[r];
            }
        }
        bf16_t* drow = dqkv + ((long)view * ATT_T + key) * rs3 + head * ATT_D;
        attb_store_t(drow + E, dk, hf);
        attb_store_t(drow + 2 * E, dv, hf);
        lds_barrier();                                       // the images are rewritten by the next block's DMA
    }
    flush_cs();
    __syncthreads();
    if (bias_ws)
        for (int i = threadIdx.x; i < E; i += ATTB_THREADS) bias_ws[(long)blockIdx.x * E + i] = cs[i];
}

}  // namespace ccd
