// mlp_bwd.h - the data-gradient chain of a transformer block's MLP branch in ONE kernel (the backward of mlp_fused.h;
// vision_transformer.py:59-65 inside Block.forward :107-113, autograd order):
//     dh  = gb . W2                      gb = bf16(g * DropPath scale): the gradient entering the branch, W2 = fc2.weight [E, H]
//     du  = dh * gelu'(u) ,  gact = gelu(u)            u = the bf16 pre-activation the forward stored
//     dy2 = du . W1                      W1 = fc1.weight [H, E]
//     g  += LayerNorm2'(dy2) ; dgamma2 / dbeta2 ; gb' = bf16(g * rowscale) ; dbias(proj) += colsum(gb') ; db1 += colsum(du)
// Round 2 ran this as two launches - the gelu'(u) product on the 256 x 256 tile (reads gb, u; writes du, gact) and the
// row-owner LayerNorm-backward product (reads du again, then x, g) - 0.34 + 0.30 ms per block.  Here du is consumed from the
// registers it was computed in: the kernel is mlp_fused.h's main loop with the two weight matrices swapped
//     P1(c): dH^T[64 hidden][32 rows] = W2^T[chunk c] . gb^T      (A = rows of fc2.weight^T, the transposed bf16 mirror)
//     P2(c): dY2^T[E][32 rows]      += W1^T[:, chunk c] . dU      (A = rows of fc1.weight^T)
// the element-wise step between them reading u (requested one chunk ahead, 16 bytes per lane and k-step) and a table of
// (Phi, gelu') PAIRS over bf16 magnitudes (one 8-byte LDS gather per element: gelu(u) = |u| Phi(|u|) + min(u, 0),
// gelu'(-u) = 1 - gelu'(u)), and rowgemm.h's LayerNorm-backward epilogue on the 32 x E accumulators.  du and gact still go to
// HBM once each - the weight-gradient products dW2 = gb^T . gact, dW1 = du^T . y2 contract over ROWS and cannot live in a
// row-owner kernel - but du is not read back, and the two HBM-bound epilogues of the old pair are one.
// Column sums of du (fc1.bias gradient): DPP halving tree per 16-register tile into a per-workgroup LDS vector, written
// once per workgroup as a row of partial sums (plain stores) that qkv_bias_finish_kernel's row-sum branch adds up.
#pragma once

namespace ccd {

struct MlpBwdParams {
    const bf16_t* gb;       // [M, E] bf16
    long ld_gb_in;
    const bf16_t* w2t;      // fc2.weight^T [H, E] bf16
    long ld2;
    const bf16_t* w1t;      // fc1.weight^T [E, H] bf16
    long ld1;
    const bf16_t* u;        // [M, H] bf16 pre-activation
    long ldu;
    bf16_t* du;             // [M, H] bf16 out
    long lddu;
    bf16_t* gact;           // [M, H] bf16 out: gelu(u)
    long ldga;
    float* db1_ws;          // [gridDim.x, H] fp32 out: per-workgroup column sums of du
    // LayerNorm-2 backward (rowgemm.h, RG_LNBWD)
    const float* x;         // x_mid [M, E] fp32
    long ldx;
    const float* mean;
    const float* rstd;
    const float* gamma;
    float* g;               // [M, E] fp32 gradient of the residual stream, in / out
    long ldg;
    int accumulate;
    float* dgamma;
    float* dbeta;
    bf16_t* gb_out;         // [M, E] bf16 = g_new * rowscale (may not alias gb: the weight-gradient launch still reads that)
    long ld_gbo;
    const float* rowscale;
    int rows_per_sample;
    float* dbias;           // [E] += column sums of gb_out
    int M, H;
};

constexpr int MB_THREADS = 256, MB_BM = 128, MB_SCRATCH = 4096;
__host__ __device__ inline int mb_smem_bytes(int E, int H) {
    return mlp_slots(E) * mlp_piece_bytes(E) + 4 * MB_SCRATCH + (2 * 1536 + 3 * E + H) * 4;   // ring, scratch, (Phi, gelu') table, cs, db1
}

template <int E>
__global__ __launch_bounds__(MB_THREADS, 1) void mlp_bwd_fused_kernel(MlpBwdParams p) {
    constexpr int KT = E / 64, KJ = E / 16, NT = E / 32, NTH = NT / 2;
    constexpr int PIECE = mlp_piece_bytes(E);
    constexpr int NSLOT = mlp_slots(E);
    constexpr int AHEAD = NSLOT - 1;
    constexpr int DEPTH = 6;
    static_assert(E % 128 == 0 && AHEAD >= 2, "ring bookkeeping");
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, hf = lane >> 5, lq = lane & 31;
    const int w = uniform_i32(t >> 6);
    char* scratch = smem + NSLOT * PIECE + w * MB_SCRATCH;
    float* lut = reinterpret_cast<float*>(smem + NSLOT * PIECE + 4 * MB_SCRATCH);      // [1536][2]: Phi(m), gelu'(m)
    float* cs = lut + 2 * (MLP_LUT_HI - MLP_LUT_LO);     // [3][E]: dgamma, dbeta, dbias of this workgroup
    float* csh = cs + 3 * E;                             // [H]: column sums of du of this workgroup
    for (unsigned i = t; i < MLP_LUT_HI - MLP_LUT_LO; i += MB_THREADS) {
        const float m = bf2f((bf16_t)(MLP_LUT_LO + i));
        const GeluTerms gt = gelu_terms(m);
        lut[2 * i] = gt.cdf;
        lut[2 * i + 1] = fmaf(m * 0.3989422804014327f, gt.gauss, gt.cdf);
    }
    for (int i = t; i < 3 * E + p.H; i += MB_THREADS) cs[i] = 0.f;
    __syncthreads();

    const int NC = p.H / 64, NP = 4 * NC;
    const int tiles = (p.M + MB_BM - 1) / MB_BM, G = gridDim.x;

    // ---- weight ring: mlp_fused.h's, with W1 := fc2.weight^T [H, E] and W2 := fc1.weight^T [E, H]
    const int dr = lane >> 3, dp = lane & 7;
    const int drow = 8 * w + dr;
    const unsigned drow2 = (unsigned)(2 * drow), swz16 = (unsigned)((dp ^ mlp_swz(drow)) * 16);
    int slot_i = 0, slot_c = 0, pos_i = 0;
    const char* req_base = nullptr;
    long req_step_a = 0, req_step_b = 0;
    unsigned req_lane = 0;
    char* req_lds = nullptr;
    auto issue_prepare = [&]() {
        int is_p2, chunk, half;
        if (pos_i < 2) { is_p2 = 0; chunk = 0; half = pos_i; }
        else {
            const int q = pos_i - 2, grp = q >> 2, r = q & 3;
            if (grp < NC - 1) { is_p2 = r >> 1; chunk = is_p2 ? grp : grp + 1; half = r & 1; }
            else { is_p2 = 1; chunk = NC - 1; half = r; }
        }
        if (!is_p2) {
            req_base = reinterpret_cast<const char*>(p.w2t) + ((long)(64 * chunk) * p.ld2 + half * (E / 2)) * 2;
            req_step_a = 64 * p.ld2;
            req_step_b = 128;
            req_lane = drow2 * (unsigned)p.ld2 + swz16;
        } else {
            req_base = reinterpret_cast<const char*>(p.w1t) + ((long)(half * (E / 2)) * p.ld1 + 64 * chunk) * 2;
            req_step_a = 64 * p.ld1;
            req_step_b = 128 * p.ld1;
            req_lane = drow2 * (unsigned)p.ld1 + swz16;
        }
        req_lds = smem + slot_i * PIECE + w * 1024;
        slot_i = slot_i + 1 == NSLOT ? 0 : slot_i + 1;
        pos_i = pos_i + 1 == NP ? 0 : pos_i + 1;
    };
    auto issue_one = [&](int i) {
        glds16(req_base + ((i & 1) * req_step_a + (i >> 1) * req_step_b) + req_lane, req_lds + 4096 * i);
    };
    const unsigned smem_addr = lds_addr_of(smem);
    auto acquire = [&]() -> unsigned {
        glds_wait<(AHEAD - 1) * KT>();
        lds_barrier();
        issue_prepare();
        const unsigned sb = smem_addr + (unsigned)(slot_c * PIECE);
        slot_c = slot_c + 1 == NSLOT ? 0 : slot_c + 1;
        return sb;
    };
    auto dma_slot = [&](auto K, auto NSTEPS) {
        constexpr int k = decltype(K)::value, stride = decltype(NSTEPS)::value / KT;
        if constexpr (k % stride == 1 && k / stride < KT) issue_one(k / stride);
    };
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
        issue_prepare();
#pragma unroll
        for (int i = 0; i < KT; ++i) issue_one(i);
    }
    // fragment read offsets inside a piece are recomputed from the lane id per piece (frag_off): held in 8 registers across the
    // chunk loop they were what the allocator spilled, and a scratch reload inside the loop waits for every older DMA piece
    auto frag_off = [&](int p2, int kk) -> unsigned {
        const int l = opaque_vgpr(t) & 63, q = l & 31, h = l >> 5;
        const int rowi = p2 ? q : ((q & 19) | ((q & 4) << 1) | ((q & 8) >> 1));     // P1: bits 2 and 3 of the weight row swapped
        return (unsigned)(rowi * 128 + (((2 * kk + h) ^ mlp_swz(rowi)) * 16));
    };
    const float inv_e = 1.0f / (float)E;
    const buf_rsrc rs_a = make_rsrc(p.gb, (unsigned)((((long)p.M - 1) * p.ld_gb_in + E) * 2));
    const buf_rsrc rs_u = make_rsrc(p.u, (unsigned)((((long)p.M - 1) * p.ldu + p.H) * 2));
    const buf_rsrc rs_du = make_rsrc(p.du, (unsigned)((((long)p.M - 1) * p.lddu + p.H) * 2));
    const buf_rsrc rs_ga = make_rsrc(p.gact, (unsigned)((((long)p.M - 1) * p.ldga + p.H) * 2));
    const buf_rsrc rs_x = make_rsrc(p.x, (unsigned)((((long)p.M - 1) * p.ldx + E) * 4));
    const buf_rsrc rs_g = make_rsrc(p.g, (unsigned)((((long)p.M - 1) * p.ldg + E) * 4));
    const buf_rsrc rs_b = make_rsrc(p.gb_out, p.gb_out ? (unsigned)((((long)p.M - 1) * p.ld_gbo + E) * 2) : 0u);
    struct LaneOff {
        int lane, hf, lq, dr, dp;
        __device__ __forceinline__ explicit LaneOff(int t) {
            lane = opaque_vgpr(t) & 63; hf = lane >> 5; lq = lane & 31; dr = lane >> 3; dp = lane & 7;
        }
        __device__ __forceinline__ unsigned frag(long ld, int elt, int per_hf) const { return (unsigned)((lq * ld + per_hf * hf) * elt); }
        __device__ __forceinline__ unsigned rows8(long ld, int elt) const { return (unsigned)(dr * ld * elt + dp * 16); }
        __device__ __forceinline__ unsigned scr_rd(int i) const { return (unsigned)((dr + 8 * i) * 128 + ((dp ^ dr) * 16)); }
        __device__ __forceinline__ unsigned scr_wr(int slot16) const { return (unsigned)(lq * 128 + ((slot16 ^ (lq & 7)) * 16)); }
    };
    const unsigned lut_addr = lds_addr_of(lut) - 8u * MLP_LUT_LO;        // byte address of the pair "magnitude 0"

    for (int tile = blockIdx.x; tile < tiles; tile += G) {
        const int m0 = tile * MB_BM, r0 = m0 + 32 * w;
        const int row = r0 + lq, grow = row < p.M ? row : p.M - 1;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        {
            bf16x8 yf[KJ];                 // this lane's row of gb as B operands of the first product
            {
                const unsigned so = (unsigned)r0 * (unsigned)(p.ld_gb_in * 2), lo_y = LaneOff(t).frag(p.ld_gb_in, 2, 8);
#pragma unroll
                for (int j = 0; j < KJ; ++j) yf[j] = __builtin_bit_cast(bf16x8, buf_load16(rs_a, lo_y, so + 32 * j));
            }
            // u of a chunk: this lane's row, the 8 hidden units of every (tile tt, k-step s) it holds in its accumulators
            // (columns 64 c + 32 tt + 16 s + 8 hf .. + 7 = one 16-byte load each), requested one chunk ahead
            u32x4 ucur[4];                 // (one set: the next chunk's is requested as soon as this chunk's last pair is finished -
                                           // a second set costs 16 registers the kernel does not have)
            auto load_u = [&](int c) {
                const unsigned so = (unsigned)r0 * (unsigned)(p.ldu * 2) + 128u * (unsigned)c, lo_u = LaneOff(t).frag(p.ldu, 2, 8);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) ucur[k4] = buf_load16(rs_u, lo_u, so + 32 * k4);
            };
            load_u(0);
            f32x16 h[2];
            u32x4 hbw[4];                  // du of the chunk being consumed, as packed bf16 B operands of the second product
            auto p1_piece = [&](auto KH, auto extra, auto filler) {
                constexpr int kh = decltype(KH)::value;
                using Extra = decltype(extra);
                const unsigned sb = acquire();
                const unsigned areg[4] = {sb + frag_off(0, 0), sb + frag_off(0, 1), sb + frag_off(0, 2), sb + frag_off(0, 3)};
                if constexpr (kh == 0) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) h[tt][r] = 0.f;
                }
                mlp_product<KJ, DEPTH, MlpMapP1<KT / 2>, Extra>(
                    areg,
                    [&](auto K, const bf16x8& a) {
                        constexpr int k = decltype(K)::value;
                        h[k & 1] = mfma_32x32x16_bf16(a, yf[(KJ / 2) * kh + (k >> 1)], h[k & 1]);
                    },
                    [&](auto K) {
                        dma_slot(K, std::integral_constant<int, KJ>{});
                        filler(K);
                    });
            };
            auto p2_piece = [&](auto HH) {
                constexpr int hh = decltype(HH)::value;
                const unsigned sb = acquire();
                const unsigned areg[4] = {sb + frag_off(1, 0), sb + frag_off(1, 1), sb + frag_off(1, 2), sb + frag_off(1, 3)};
                mlp_product<4 * NTH, DEPTH, MlpMapP2<NTH>, MlpNoExtra>(
                    areg,
                    [&](auto K, const bf16x8& a) {
                        constexpr int k = decltype(K)::value;
                        acc[NTH * hh + k % NTH] = mfma_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, hbw[k / NTH]), acc[NTH * hh + k % NTH]);
                    },
                    [&](auto K) { dma_slot(K, std::integral_constant<int, 4 * NTH>{}); });
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            p1_piece(I0{}, MlpNoExtra{}, [](auto) {});
            p1_piece(I1{}, MlpNoExtra{}, [](auto) {});
#pragma unroll 1
            for (int c = 0; c < NC; ++c) {
                // element-wise step of chunk c, two elements ("a pair") at a time: registers 8 s + (0 .. 7) of tile tt are hidden
                // units 64 c + 32 tt + 16 s + 8 hf + (0 .. 7); dword e of uq[2 tt + s] holds u of units 2 e, 2 e + 1 of those 8.
                // Issued (two 8-byte gathers of (Phi, gelu') by magnitude) at one MFMA step of the NEXT chunk's first product,
                // finished LAG steps later - mlp_fused.h's schedule.
                f32x16 hv[2] = {h[0], h[1]};   // dh; overwritten by du (fp32) pair by pair for the column sums
                // table values of the pairs in flight: a pair is finished LAG = 7 steps after it was issued; issued every S steps
                constexpr int TS = MlpGeluSchedule<KJ, DEPTH>::S >= 2 ? 4 : 8;
                lds_f32x2 t0[TS], t1[TS];
                // du and gelu(u) leave through the wave's scratch as two HALF images per tile tt: [32 rows][32 hidden] bf16 of du in
                // its first 2 KiB, of gelu(u) in the second - both written by the pair that completes a 16-byte slot, flushed as
                // 64-byte row segments when the tile's eight pairs are done (holding gelu(u)'s words of a whole chunk in registers
                // until du's image had been read cost 16 registers: 63 spilled)
                auto pair_issue = [&](auto PI) {
                    constexpr int pi = decltype(PI)::value, k4 = pi >> 2, e = pi & 3, sl = pi & (TS - 1);
                    const unsigned uw = ucur[k4][e];
                    unsigned m0_ = uw & 0x7fffu, m1_ = (uw >> 16) & 0x7fffu;
                    m0_ = m0_ < MLP_LUT_LO ? MLP_LUT_LO : (m0_ > MLP_LUT_HI - 1u ? MLP_LUT_HI - 1u : m0_);
                    m1_ = m1_ < MLP_LUT_LO ? MLP_LUT_LO : (m1_ > MLP_LUT_HI - 1u ? MLP_LUT_HI - 1u : m1_);
                    lds_gather_f32x2(t0[sl], lut_addr + 8u * m0_);
                    lds_gather_f32x2(t1[sl], lut_addr + 8u * m1_);
                };
                u32x4 gw;                      // gelu(u) words of the 16-byte slot being completed
                auto pair_finish = [&](auto PI) {
                    constexpr int pi = decltype(PI)::value, k4 = pi >> 2, e = pi & 3, r = 8 * (k4 & 1) + 2 * e, sl = pi & (TS - 1);
                    lds_landed(t0[sl], t1[sl]);
                    const unsigned uw = ucur[k4][e];
                    const float u0 = bf_lo(uw), u1 = bf_hi(uw);
                    const float d0 = u0 < 0.f ? 1.0f - t0[sl].y : t0[sl].y, d1 = u1 < 0.f ? 1.0f - t1[sl].y : t1[sl].y;
                    const float du0 = hv[k4 >> 1][r] * d0, du1 = hv[k4 >> 1][r + 1] * d1;
                    hv[k4 >> 1][r] = du0;
                    hv[k4 >> 1][r + 1] = du1;
                    hbw[k4][e] = pack_bf2(du0, du1);
                    gw[e] = pack_bf2(fmaf(fabsf(u0), t0[sl].x, fminf(u0, 0.f)), fmaf(fabsf(u1), t1[sl].x, fminf(u1, 0.f)));
                    if (e == 3) {          // slot (k-step s = k4 & 1, half hf) of the tile's two half images
                        const LaneOff lo(t);
                        const unsigned at = (unsigned)(lo.lq * 64 + (((2 * (k4 & 1) + lo.hf) ^ (lo.lq & 3)) * 16));
                        *reinterpret_cast<u32x4*>(scratch + at) = hbw[k4];
                        *reinterpret_cast<u32x4*>(scratch + 2048 + at) = gw;
                    }
                };
                auto flush_tile = [&](int tt) {      // the two half images of tile tt -> du / gact [rows][64 c + 32 tt .. + 31]
                    wave_lds_fence();
                    const LaneOff lo(t);
                    const int fr_ = lo.lane >> 2, fp = lo.lane & 3;                       // 16 rows x 4 slots per instruction
                    const unsigned lo_d = (unsigned)(fr_ * p.lddu * 2 + fp * 16), lo_g = (unsigned)(fr_ * p.ldga * 2 + fp * 16);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned at = (unsigned)((fr_ + 16 * i) * 64 + ((fp ^ (fr_ & 3)) * 16));
                        const u32x4 vd = *reinterpret_cast<const u32x4*>(scratch + at);
                        const u32x4 vg = *reinterpret_cast<const u32x4*>(scratch + 2048 + at);
                        buf_store16(rs_du, lo_d, (unsigned)(r0 + 16 * i) * (unsigned)(p.lddu * 2) + 128 * c + 64 * tt, vd);
                        buf_store16(rs_ga, lo_g, (unsigned)(r0 + 16 * i) * (unsigned)(p.ldga * 2) + 128 * c + 64 * tt, vg);
                    }
                    wave_lds_fence();
                };
                if (c + 1 < NC) {
                    auto piece_with_gelu = [&](auto KH) {
                        constexpr int kh = decltype(KH)::value;
                        p1_piece(KH, MlpGeluSchedule<KJ, DEPTH>{}, [&](auto K) {
                            constexpr int k = decltype(K)::value;
                            using Sch = MlpGeluSchedule<KJ, DEPTH>;
                            if constexpr (k >= Sch::LAG && (k - Sch::LAG) % Sch::S == 0 && (k - Sch::LAG) / Sch::S < 8)
                                pair_finish(std::integral_constant<int, 8 * kh + (k - Sch::LAG) / Sch::S>{});
                            if constexpr (Sch::at(k) != 0) pair_issue(std::integral_constant<int, 8 * kh + k / Sch::S>{});
                        });
                        constexpr int first_late = (KJ - 1 - MlpGeluSchedule<KJ, DEPTH>::LAG) / MlpGeluSchedule<KJ, DEPTH>::S + 1;
                        if constexpr (first_late < 8) {
                            lds_drain();
                            mlp_static_for<8 * kh + (first_late < 0 ? 0 : first_late), 8 * kh + 8>(pair_finish);
                        }
                    };
                    piece_with_gelu(I0{});
                    flush_tile(0);
                    piece_with_gelu(I1{});
                    load_u(c + 1);                 // this chunk's u is used up: the next chunk's arrives under the second product
                    flush_tile(1);
                } else {
                    mlp_static_for<0, 4>([&](auto Gq) {
                        constexpr int gq_ = decltype(Gq)::value;
                        mlp_static_for<4 * gq_, 4 * gq_ + 4>(pair_issue);
                        lds_drain();
                        mlp_static_for<4 * gq_, 4 * gq_ + 4>(pair_finish);
                        if (gq_ & 1) flush_tile(gq_ >> 1);
                    });
                }
                // fc1.bias gradient: column sums of du over the wave's 32 rows (register r of tile tt = hidden unit
                // 64 c + 32 tt + 16 (r >> 3) + 8 hf + (r & 7)); rows beyond M hold gb = 0 and contribute 0
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = hv[tt][r];
                    const float tot = rg_fold16(v, lq);
                    atomicAdd(csh + 64 * c + 32 * tt + 16 * ((lq >> 3) & 1) + 8 * hf + (lq & 7), tot);
                }
                p2_piece(I0{});
                p2_piece(I1{});
            }
        }
        // ---- epilogue: LayerNorm-2 backward on acc = dy2 (rowgemm.h's RG_LNBWD passes; rows complete inside lanes l, l ^ 32)
        const float mu = p.mean[grow], rs = p.rstd[grow];
        const unsigned so_x = (unsigned)r0 * (unsigned)(p.ldx * 4);
        float s1 = 0.f, sq = 0.f;
        {
            const unsigned lo_x = LaneOff(t).frag(p.ldx, 4, 4);
            constexpr int PA = 4;
            u32x4 xb[PA][4];
            auto load_x = [&](int nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) xb[nt % PA][g] = buf_load16(rs_x, lo_x, so_x + (32 * nt + 8 * g) * 4);
            };
#pragma unroll
            for (int nt = 0; nt < PA - 1 && nt < NT; ++nt) load_x(nt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt + PA - 1 < NT) load_x(nt + PA - 1);
                float vg[16], vb[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4v x = __builtin_bit_cast(f32x4v, xb[nt % PA][g]);
                    const f32x4v ga = *reinterpret_cast<const f32x4v*>(p.gamma + 32 * nt + 8 * g + 4 * opaque_vgpr(hf));
                    const float xx[4] = {x.x, x.y, x.z, x.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dy = acc[nt][4 * g + e], xh = (xx[e] - mu) * rs, dg = dy * gg[e];
                        acc[nt][4 * g + e] = opaque_f32(__builtin_bit_cast(float, rg_pack(dg, xh)));
                        s1 += dg;
                        sq = fmaf(dg, xh, sq);
                        vg[4 * g + e] = dy * xh;
                        vb[4 * g + e] = dy;
                    }
                }
                rg_colsum16(vg, cs + 32 * nt, lq, hf);
                rg_colsum16(vb, cs + E + 32 * nt, lq, hf);
                CCD_SCHED_FENCE();
                asm volatile("" ::: "memory");
            }
        }
        s1 += shfl_xor(s1, 32);
        sq += shfl_xor(sq, 32);
        const float c1 = rs * s1 * inv_e, c2 = rs * sq * inv_e;
        float sc = 1.0f;
        if (p.gb_out && p.rowscale) sc = p.rowscale[grow / p.rows_per_sample];
        {
            const LaneOff lo(t);
            const unsigned lo_gl = lo.frag(p.ldg, 4, 4), so_g = (unsigned)r0 * (unsigned)(p.ldg * 4);
            const unsigned lo_o = lo.rows8(p.ldg, 4), lo_n = lo.rows8(p.ld_gbo, 2);
            constexpr int PB = 3;
            u32x4 gbuf[PB][4];
            auto load_xg = [&](int nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    gbuf[nt % PB][g] = u32x4{0u, 0u, 0u, 0u};
                    if (p.accumulate) gbuf[nt % PB][g] = buf_load16(rs_g, lo_gl, so_g + (32 * nt + 8 * g) * 4);
                }
            };
#pragma unroll
            for (int nt = 0; nt < PB - 1 && nt < NT; ++nt) load_xg(nt);
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                u32x2 ypk[2][4];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int nt = 2 * np + tt;
                    if (nt + PB - 1 < NT) load_xg(nt + PB - 1);
                    float vbi[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4v go = __builtin_bit_cast(f32x4v, gbuf[nt % PB][g]);
                        const float oo[4] = {go.x, go.y, go.z, go.w};
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float pf = acc[nt][4 * g + e];
                            const unsigned pk = __builtin_bit_cast(unsigned, pf);
                            const float dx = fmaf(-c2, rg_unpack_xh(pk), fmaf(rg_unpack_dg(pk), rs, -c1));
                            v[e] = oo[e] + dx;
                        }
                        *reinterpret_cast<f32x4v*>(scratch + lo.scr_wr(2 * g + lo.hf)) = f32x4v{v[0], v[1], v[2], v[3]};
                        if (p.gb_out) {
                            ypk[tt][g].x = pack_bf2(v[0] * sc, v[1] * sc);
                            ypk[tt][g].y = pack_bf2(v[2] * sc, v[3] * sc);
                            vbi[4 * g] = bf_lo(ypk[tt][g].x); vbi[4 * g + 1] = bf_hi(ypk[tt][g].x);
                            vbi[4 * g + 2] = bf_lo(ypk[tt][g].y); vbi[4 * g + 3] = bf_hi(ypk[tt][g].y);
                        }
                    }
                    if (p.gb_out && p.dbias) rg_colsum16(vbi, cs + 2 * E + 32 * nt, lq, hf);
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                        buf_store16(rs_g, lo_o, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldg * 4) + 128 * nt, o);
                    }
                    wave_lds_fence();
                    CCD_SCHED_FENCE();
                    asm volatile("" ::: "memory");
                }
                if (p.gb_out) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<u32x2*>(scratch + lo.scr_wr(4 * tt + g) + 8 * lo.hf) = ypk[tt][g];
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                        buf_store16(rs_b, lo_n, (unsigned)(r0 + 8 * i) * (unsigned)(p.ld_gbo * 2) + 128 * np, o);
                    }
                    wave_lds_fence();
                }
            }
        }
    }
    glds_wait_all();
    __syncthreads();
    for (int i = t; i < E; i += MB_THREADS) {
        atomicAdd(p.dgamma + i, cs[i]);
        atomicAdd(p.dbeta + i, cs[E + i]);
        if (p.dbias) atomicAdd(p.dbias + i, cs[2 * E + i]);
    }
    for (int i = t; i < p.H; i += MB_THREADS) p.db1_ws[(long)blockIdx.x * p.H + i] = csh[i];
}

}  // namespace ccd
