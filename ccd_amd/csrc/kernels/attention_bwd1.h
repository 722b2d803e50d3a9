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
constexpr int ATTB1_RING = 9 * 4096;                                     // Q / dO row images as rings of nine 32-row tile slots
constexpr int ATTB1_SMEM = 2 * ATTB1_RING + ATTB_IMG + 2 * ATTB1_XBUF + 4 * ATT_T * 4 + ATTB_CS_BYTES;
// 8-byte slot swizzle of an exchange row (32 queries = 8 slots): writes (16 keys x one slot) and transposing reads (4 keys x 4 slots
// per 16-lane group, two groups per LDS cycle) both touch every bank once
__device__ __forceinline__ int attb1_fx(int key) { return ((key >> 2) & 3) | ((key >> 1) & 4); }

// Cross-block prefetch without a second set of images (they would not fit): the Q / dO images are RINGS of nine tile slots - block
// number `it` of a workgroup keeps tile qt in slot (8 it + qt) % 9, so one slot is always free: the next block's tile 0 goes there at
// once and tile qt + 1 into the slot tile qt leaves behind the step's barrier (one 1-KiB DMA piece per wave and step).  The K image
// is dead once the block's fragments are in registers: the next block's K rows land in it during the steps.  The next block's v rows
// and o rows (delta) are requested into the registers of this block's k / v fragments right behind the last score products.  All
// of it has landed when the last dQ product is done (`vmcnt(0)` in front of the block's LAST stores, which then drain under the
// next block's start).
__global__ __launch_bounds__(512) void attention_bwd_onepass_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                    const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                    bf16_t* __restrict__ dqkv, float* __restrict__ bias_ws,
                                                                    int heads, float scale, int nblocks, int lab) {
    char* smem = dynamic_smem();
    char* q_ring = smem;
    char* do_ring = smem + ATTB1_RING;
    char* k_img = smem + 2 * ATTB1_RING;
    char* x_buf = k_img + ATTB_IMG;                                      // [2][8 waves][32 keys][64 B]
    float* lse_s = reinterpret_cast<float*>(x_buf + 2 * ATTB1_XBUF);     // [2][256]: this block's and the next one's
    float* del_s = lse_s + 2 * ATT_T;
    float* lse_sw = del_s + ATT_T;
    float* cs = lse_sw + ATT_T;                                           // [heads][64] column sums of dQ over this workgroup's blocks
    const int lane = threadIdx.x & 63, hf = lane >> 5, lq = lane & 31;
    const int w = uniform_i32(threadIdx.x >> 6);             // wave index as a scalar: what derives from it stays in SGPRs
    const int E = heads * ATT_D;
    const long rs3 = 3L * E;
    const int key = 32 * w + lq;
    const int d16 = w & 3, q16 = w >> 2, g4 = lane >> 4, l16 = lane & 15;
    if (bias_ws)
        for (int i = threadIdx.x; i < E; i += ATTB_THREADS) cs[i] = 0.f;    // (published by the first block's barrier)

    // ---- requests (1-KiB LDS-DMA piece = 8 rows x 128 B; lane L fetches the 16 bytes the swizzle puts at row L / 8, position L % 8)
    const int views = nblocks / heads;                       // block b = (head b / views, view b % views): a workgroup's consecutive blocks
                                                             // share the head, its dQ column sums are flushed once per head
    auto dma_tile = [&](int blk, int qt, int slot) {         // tile qt of Q (waves 0-3: 8 rows each) and of dO (waves 4-7)
        const int view = blk % views, head = blk / views;
        const int ln = opaque_vgpr((int)threadIdx.x) & 63;
        const int r = 8 * (w & 3) + (ln >> 3), row = 32 * qt + r, src = (ln & 7) ^ attb_swz2(r);
        if (w < 4) glds16(qkv + ((long)view * ATT_T + row) * rs3 + head * ATT_D + src * 8, q_ring + slot * 4096 + (w & 3) * 1024);
        else glds16(d_o + ((long)view * ATT_T + row) * E + head * ATT_D + src * 8, do_ring + slot * 4096 + (w & 3) * 1024);
    };
    auto dma_k_lse = [&](int blk, int par) {                 // the K rows (4 pieces per wave) and lse of a block
        const int view = blk % views, head = blk / views;
        const bf16_t* k_base = qkv + (long)view * ATT_T * rs3 + head * ATT_D + E;
        const int ln = opaque_vgpr((int)threadIdx.x) & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (4 * w + i) + (ln >> 3), src = (ln & 7) ^ attb_swz2(row);
            glds16(k_base + (unsigned)(row * (int)rs3 + src * 8), k_img + (4 * w + i) * 1024);
        }
        if (w < 4) glds4(lse + ((long)view * heads + head) * ATT_T + 64 * w + ln, reinterpret_cast<char*>(lse_s + par * ATT_T) + 256 * w);
    };
    // kfw: this wave's k fragments - and, between a block's last score products and the next block's delta, the o rows of its 32
    // queries (index 32 w + lq); vfw: its v fragments = the v rows as they lie in memory.  One register set for both lives.
    u32x4 kfw[4] = {}, vfw[4] = {};        // (defined: the tracked loads below take them as in-out operands)
    auto request_vo = [&](int blk) {
        const int view = blk % views, head = blk / views;
        // (one base pointer per tensor + immediate offsets, derived here from an opaque lane id: eight precomputed 64-bit addresses
        // held across the steps were what the register allocator spilled)
        const int ln = opaque_vgpr((int)threadIdx.x) & 63, kq = 32 * w + (ln & 31);
        const bf16_t* vrow = qkv + ((long)view * ATT_T + kq) * rs3 + head * ATT_D + 2 * E + 8 * (ln >> 5);
        const bf16_t* orow = o + ((long)view * ATT_T + kq) * E + head * ATT_D + 8 * (ln >> 5);
        global_load16_late<0>(vfw[0], vrow); global_load16_late<32>(vfw[1], vrow); global_load16_late<64>(vfw[2], vrow); global_load16_late<96>(vfw[3], vrow);
        global_load16_late<0>(kfw[0], orow); global_load16_late<32>(kfw[1], orow); global_load16_late<64>(kfw[2], orow); global_load16_late<96>(kfw[3], orow);
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
    const unsigned q_addr = smem_addr, do_addr = smem_addr + ATTB1_RING, k_addr = smem_addr + 2 * ATTB1_RING;
    const unsigned x_addr = k_addr + ATTB_IMG;
    const unsigned stat_addr = lds_addr_of(lse_sw) + (unsigned)(16 * hf);       // (delta: ATT_T floats below)
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

#ifdef CCD_ATTB1_LAB      // lab build: cycle totals of wave 0 / wave 7 per phase -> bias_ws[blockIdx.x * E + 16 (w == 7) + i] as floats
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define AB1_STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tprev; tprev = tn_; }
#else
#define AB1_STAMP(i)
#endif
    int blk = blockIdx.x;
    if (blk >= nblocks) return;
    if (w >= 4 && !(lab & 1)) wave_prio<1>();                // (guide: static priority for the younger half of an 8-wave workgroup;
                                                             // lab bit 1: none, bit 2: the older half instead)
    if (w < 4 && (lab & 2)) wave_prio<1>();
    {                                                        // the first block: everything up front
#pragma unroll
        for (int qt = 0; qt < 8; ++qt) dma_tile(blk, qt, qt);
        dma_k_lse(blk, 0);
        request_vo(blk);
        glds_wait_all();
    }
    int b0 = 0;                                              // ring slot of this block's tile 0
    for (int it = 0; blk < nblocks; blk += gridDim.x, ++it) {
        const int view = blk % views, head = blk / views, nxt = blk + (int)gridDim.x;
        const bool more = nxt < nblocks;
        if (head != cs_head) { flush_cs(); cs_head = head; }
        const float* lse_b = lse_s + (it & 1) * ATT_T;       // as the DMA wrote it; lse_sw: scaled for exp2
        AB1_STAMP(0)
        lds_barrier();                                       // this block's images are complete (every wave drained its requests) and
                                                             // every wave has left the previous block
        AB1_STAMP(1)
        // ---- per block: k / v fragments of this wave's keys, delta of its 32 queries, the K^T fragments of its 16 d columns
        bf16x8 kT[8];
        vm_landed4(vfw);
        vm_landed4(kfw);
        {
            const int f = attb_swz2(key);
            const int dslot = b0 + w >= 9 ? b0 + w - 9 : b0 + w;        // dO rows of this wave's queries: tile w
            float dsum = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const u32x4 dw = *reinterpret_cast<const u32x4*>(do_ring + dslot * 4096 + lq * 128 + (((2 * kk + hf) ^ attb_swz2(lq)) << 4));
                float a[8], b[8];
                unpack8(dw, a);
                unpack8(kfw[kk], b);                         // (the o rows)
#pragma unroll
                for (int e = 0; e < 8; ++e) dsum += a[e] * b[e];
            }
            dsum += shfl_xor(dsum, 32);
            if (hf == 0) del_s[key] = dsum;
            // lse -> - lse * log2(e), once per block instead of once per score (the softmax below is the block's largest VALU item)
            if (threadIdx.x < ATT_T) lse_sw[threadIdx.x] = lse_b[threadIdx.x] * -1.4426950408889634f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) kfw[kk] = *reinterpret_cast<const u32x4*>(k_img + key * 128 + (((2 * kk + hf) ^ f) << 4));
            tr_u32x2 tl[8], th[8];
            unsigned koff[2];                                // (from an opaque lane id: not kept in registers across the steps)
            {
                const int ln = opaque_vgpr((int)threadIdx.x) & 63, lg4 = ln >> 4, ll16 = ln & 15;
#pragma unroll
                for (int rsel = 0; rsel < 2; ++rsel) {
                    const int kk = 8 * lg4 + 4 * rsel + (ll16 >> 2), d0 = 16 * d16 + 4 * (ll16 & 3);
                    koff[rsel] = (unsigned)(kk * 128 + (((d0 >> 3) ^ attb_swz2(kk)) << 4) + (d0 & 7) * 2);  // (+ 32 ks rows: swz2 has period 16)
                }
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                lds_read_tr<0>(tl[ks], k_addr + (unsigned)(ks * 4096) + koff[0]);
                lds_read_tr<0>(th[ks], k_addr + (unsigned)(ks * 4096) + koff[1]);
            }
            lds_drain();
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                kT[ks] = frag_from_tr(tl[ks], th[ks]);
                asm volatile("" : "+v"(kT[ks]));
            }
        }
        AB1_STAMP(2)
        lds_barrier();                                       // delta is published; every wave is done with the K image
        AB1_STAMP(1)
        if (more) {
            dma_k_lse(nxt, (it + 1) & 1);
            dma_tile(nxt, 0, b0 + 8 >= 9 ? b0 - 1 : b0 + 8);
        }

        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
#pragma unroll 1
        for (int qt = 0; qt < 8; ++qt) {
            const int slot = b0 + qt >= 9 ? b0 + qt - 9 : b0 + qt;
            const unsigned tile = (unsigned)(slot * 4096);
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            {
                // the eight row fragments of the step by hand, all in flight at once (left to the compiler: read two, lgkmcnt(0),
                // two products, four LDS round trips in a row - the score products were 22 % of a block)
                const int f = attb_swz2(lq);
                bf16x8 qa[4], da[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const unsigned off = tile + (unsigned)(lq * 128 + (((2 * kk + hf) ^ f) << 4));
                    lds_read_frag<0>(qa[kk], q_addr + off);
                    lds_read_frag<0>(da[kk], do_addr + off);
                }
                // The step's statistics (this lane's 16 queries: 4 x 16 bytes of - lse log2 e, 4 x 16 of delta) ride in the registers the
                // first / last four fragments leave: requested between the products, they land under them instead of as four LDS round
                // trips in front of the exponentials (left to the compiler the softmax was 30 % of a block).
                auto stat_read = [&](bf16x8& dst, int j, int which) {
                    lds_read_frag<0>(dst, stat_addr + (unsigned)(qt * 128 + j * 32) - (unsigned)(which * ATT_T * 4));
                };
                lds_wait_frag<7>(qa[0]);
                s = mfma_32x32x16_bf16(qa[0], __builtin_bit_cast(bf16x8, kfw[0]), s);           // S[q][key]
                lds_wait_frag<6>(da[0]);
                dp = mfma_32x32x16_bf16(da[0], __builtin_bit_cast(bf16x8, vfw[0]), dp);         // dP[q][key]
                lds_wait_frag<5>(qa[1]);
                s = mfma_32x32x16_bf16(qa[1], __builtin_bit_cast(bf16x8, kfw[1]), s);
                lds_wait_frag<4>(da[1]);
                dp = mfma_32x32x16_bf16(da[1], __builtin_bit_cast(bf16x8, vfw[1]), dp);
                stat_read(qa[0], 0, 0); stat_read(da[0], 1, 0); stat_read(qa[1], 2, 0); stat_read(da[1], 3, 0);
                lds_wait_frag<7>(qa[2]);
                s = mfma_32x32x16_bf16(qa[2], __builtin_bit_cast(bf16x8, kfw[2]), s);
                lds_wait_frag<6>(da[2]);
                dp = mfma_32x32x16_bf16(da[2], __builtin_bit_cast(bf16x8, vfw[2]), dp);
                lds_wait_frag<5>(qa[3]);
                s = mfma_32x32x16_bf16(qa[3], __builtin_bit_cast(bf16x8, kfw[3]), s);
                lds_wait_frag<4>(da[3]);
                dp = mfma_32x32x16_bf16(da[3], __builtin_bit_cast(bf16x8, vfw[3]), dp);
                stat_read(qa[2], 0, 1); stat_read(da[2], 1, 1); stat_read(qa[3], 2, 1); stat_read(da[3], 3, 1);
                if (qt == 7 && more) request_vo(nxt);        // (the k / v fragments have fed their last products)
                AB1_STAMP(3)
                // p = exp2(S scale log2 e - lse log2 e) for the 16 scores, then dS / scale = p (dP - delta): the factor goes onto dK and
                // dQ where they leave (4 + 2 x 16 values per step and block instead of 16 per step)
                bf16x8* const stl[4] = {&qa[0], &da[0], &qa[1], &da[1]};
                bf16x8* const std_[4] = {&qa[2], &da[2], &qa[3], &da[3]};
                mlp_static_for<0, 4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    lds_wait_frag<7 - j>(*stl[j]);
                    const f32x4v lv = __builtin_bit_cast(f32x4v, *stl[j]);
                    const float l4[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[4 * j + i] = fast_exp2(fmaf(s[4 * j + i], scale * 1.4426950408889634f, l4[i]));
                });
                mlp_static_for<0, 4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    lds_wait_frag<3 - j>(*std_[j]);
                    const f32x4v dl = __builtin_bit_cast(f32x4v, *std_[j]);
                    const float d4[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) dp[4 * j + i] = s[4 * j + i] * (dp[4 * j + i] - d4[i]);
                });
            }
            AB1_STAMP(4)
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
            AB1_STAMP(5)
            lds_barrier();                                   // every wave's dS tile of this step is in the exchange image, and every
                                                             // wave is done with this step's Q / dO tile (LDS-only barrier: the dQ
                                                             // stores and the prefetch requests stay in flight)
            AB1_STAMP(6)
            if (more && qt < 7) dma_tile(nxt, qt + 1, slot);
            // ---- dQ piece [16 d][16 queries] of this step over all 256 keys
            f32x4 dq = {0.f, 0.f, 0.f, 0.f};
            const unsigned xb = x_addr + (unsigned)((qt & 1) * ATTB1_XBUF);
            mlp_static_for<0, 2>([&](auto HB) {              // two batches of four key tiles (eight transposing reads in flight)
                constexpr int hb = decltype(HB)::value;
                tr_u32x2 xl[4], xh[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    lds_read_tr<0>(xl[ks], xb + (unsigned)((4 * hb + ks) * 2048) + xoff[0]);
                    lds_read_tr<0>(xh[ks], xb + (unsigned)((4 * hb + ks) * 2048) + xoff[1]);
                }
                mlp_static_for<0, 4>([&](auto KS) {
                    constexpr int ks = decltype(KS)::value;
                    bf16x8 bfr = frag_from_tr(xl[ks], xh[ks]);
                    lds_wait_frag<2 * (3 - ks)>(bfr);
                    dq = mfma_16x16x32_bf16(kT[4 * hb + ks], bfr, dq);
                });
            });
            AB1_STAMP(7)
            if (qt == 7) glds_wait_all();                    // the next block's images and rows have landed (requested at least a step ago;
                                                             // what is still in flight are the stores of earlier steps)
            {
                u32x2 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) { dq[r] *= scale; csq[r] += dq[r]; }
                pk.x = pack_bf2(dq[0], dq[1]);
                pk.y = pack_bf2(dq[2], dq[3]);
                bf16_t* dst = dqkv + ((long)view * ATT_T + 32 * qt + 16 * q16 + l16) * rs3 + head * ATT_D + 16 * d16 + 4 * g4;
                *reinterpret_cast<u32x2*>(dst) = pk;
            }
            AB1_STAMP(8)
        }
        bf16_t* drow = dqkv + ((long)view * ATT_T + key) * rs3 + head * ATT_D;
        att_store_row16(drow + E, dk, hf, scale);
        attb_store_t(drow + 2 * E, dv, hf);
        b0 = b0 + 8 >= 9 ? b0 - 1 : b0 + 8;
        AB1_STAMP(9)
    }
    flush_cs();
    __syncthreads();
    if (bias_ws)
        for (int i = threadIdx.x; i < E; i += ATTB_THREADS) bias_ws[(long)blockIdx.x * E + i] = cs[i];
#ifdef CCD_ATTB1_LAB
    __syncthreads();
    if (bias_ws && lane == 0 && (w == 0 || w == 7))
        for (int i = 0; i < 10; ++i) bias_ws[(long)blockIdx.x * E + (w == 7 ? 16 : 0) + i] = (float)ph[i];
#endif
}

}  // namespace ccd
