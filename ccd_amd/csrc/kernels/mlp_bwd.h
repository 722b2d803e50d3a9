// mlp_bwd.h - the data-gradient chain of a transformer block's MLP branch in ONE kernel: the backward of mlp_fused.h
// (vision_transformer.py:59-65 inside Block.forward :107-113, autograd order):
//     dh  = gb . W2                      gb = bf16(g * DropPath scale): the gradient entering the branch, W2 = fc2.weight [E, H]
//     du  = dh * gelu'(u)                u = the bf16 pre-activation the forward pass stored
//     dy2 = du . W1                      W1 = fc1.weight [H, E]
//     g  += LayerNorm2'(dy2) ; dgamma2 / dbeta2 ; gb' = bf16(g * rowscale) ; dbias(proj) += colsum(gb') ; db1 += colsum(du)
// The step runs this as two launches - the gelu'(u) product on the 256 x 256 tile (gemm256.h: gb, u in; du, gelu(u) out) and the
// row-owner LayerNorm-backward product (rowgemm.h: du read again, then x, g) - 0.34 + 0.30 ms per block.  Here du is consumed from
// the registers it is computed in.  Round 3 built this once (836 against 717 us then): its u rows came in as a row per lane, one
// chunk ahead, through the same in-order vmcnt queue as the weight ring - every ring wait behind a u request waited for that
// request's HBM latency.  This version differs in what that measurement pointed at:
//   * u arrives by LDS-DMA (1-KiB instructions of 8 rows x 128 contiguous bytes) into one of two 4-KiB images per wave, TWO chunks
//     (eight weight pieces) ahead of its use: the request is six windows old when the ring first has to wait behind it;
//   * gelu(u) is NOT produced here (the forward kernel stores it: its second-product operands ARE that tensor) - the element-wise
//     stage is one table gather and ~7 VALU instructions per element, the same weight as the forward's GELU;
//   * du leaves as 128-byte row segments through the image its u came in by; the residual-gradient stream is bf16 (round 6).
// The kernel is mlp_fused.h's main loop with the two weight matrices swapped
//     P1(c): dH^T[64 hidden][32 rows] = W2^T[chunk c] . gb^T      (A = rows of fc2.weight^T, the transposed bf16 mirror)
//     P2(c): dY2^T[E][32 rows]      += W1^T[:, chunk c] . dU      (A = rows of fc1.weight^T)
// on a 4-slot ring (96 KiB; the u images take the fifth slot's room), and rowgemm.h's LayerNorm-backward epilogue on the 32 x E
// accumulators.  Column sums of du (fc1.bias gradient): DPP halving tree per 16-register tile into a per-workgroup LDS vector, added
// to the gradient with one sweep of atomics per workgroup.
//
// VMEM bookkeeping (one in-order counter for ring requests, u requests and stores).  Windows of one chunk iteration c, in issue order:
//     W0 = P1(c+1)a: 6 ring requests | W1 = P1(c+1)b: 6 | F: 4 du stores | W2 = P2(c)a: 6 ring + 4 u requests (steps 3, 7, 11, 15) |
//     W3 = P2(c)b: 6
// A window needs the piece requested three windows earlier; "landed" = at most N requests younger than its last one are outstanding,
// N counting LOADS only (stores behind them only make the wait stricter; a store is never the oldest thing a wait asks for except at
// W0, two windows after F):  W1, W2: 12;  W3: 16 (W2's ten);  W0: 16 from the second chunk on (the previous W2's ten), 12 for the
// first (its predecessors are the tile's two leading P1 windows).  W2 ALWAYS issues its four u requests - chunk c + 2 of this tile,
// or chunks 0 / 1 of the workgroup's next tile from the last two chunks - so the counts do not depend on the position in the tile.
// Rows behind M are clamped to the last row, never out of range: an out-of-range request of a whole wave need not keep its place in
// the return order (tests/hipsim's late-DMA model reproduces what that does to a counted wait).
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
    float* db1;             // [H] fp32 += column sums of du
    // LayerNorm-2 backward (rowgemm.h, RG_LNBWD on the bf16 stream)
    const float* x;         // x_mid [M, E] fp32
    long ldx;
    const float* mean;
    const float* rstd;
    const float* gamma;
    bf16_t* g;              // [M, E] bf16 gradient of the residual stream, in / out
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

constexpr int MB_THREADS = 256, MB_BM = 128, MB_UBUF = 4096, MB_NSLOT = 4;
__host__ __device__ inline int mb_smem_bytes(int E, int H) {
    return MB_NSLOT * mlp_piece_bytes(E) + 4 * 2 * MB_UBUF + (1536 + 4 * E + H) * 4;   // ring, u images, gelu' table, gamma + 3 sums, db1 sums
}

template <int E>
__global__ __launch_bounds__(MB_THREADS, 1) void mlp_bwd_fused_kernel(MlpBwdParams p) {
    constexpr int KT = E / 64, KJ = E / 16, NT = E / 32, NTH = NT / 2;
    constexpr int PIECE = mlp_piece_bytes(E);
    constexpr int NSLOT = MB_NSLOT, AHEAD = NSLOT - 1;
    constexpr int DEPTH = 6;
    constexpr int WAIT_RING = (AHEAD - 1) * KT, WAIT_RING_U = WAIT_RING + 4;
    static_assert(E % 128 == 0 && AHEAD >= 2 && 4 * NTH >= 16, "ring bookkeeping; the u requests ride behind MFMA steps 3, 7, 11, 15 of a second-product window");
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, hf = lane >> 5, lq = lane & 31;
    const int w = uniform_i32(t >> 6);
    char* ubuf = smem + NSLOT * PIECE + w * (2 * MB_UBUF);             // this wave's two u / du images
    float* lut = reinterpret_cast<float*>(smem + NSLOT * PIECE + 4 * 2 * MB_UBUF);      // gelu'(m) over bf16 magnitudes
    float* vga = lut + (MLP_LUT_HI - MLP_LUT_LO);
    float* cs = vga + E;                                 // [3][E]: dgamma, dbeta, dbias of this workgroup
    float* csh = cs + 3 * E;                             // [H]: column sums of du of this workgroup
    for (unsigned i = t; i < MLP_LUT_HI - MLP_LUT_LO; i += MB_THREADS) {
        const float m = bf2f((bf16_t)(MLP_LUT_LO + i));
        const GeluTerms gt = gelu_terms(m);
        lut[i] = fmaf(m * 0.3989422804014327f, gt.gauss, gt.cdf);
    }
    for (int i = t; i < E; i += MB_THREADS) vga[i] = p.gamma[i];
    for (int i = t; i < 3 * E + p.H; i += MB_THREADS) cs[i] = 0.f;
    __syncthreads();

    const int NC = p.H / 64, NP = 4 * NC;
    const int tiles = (p.M + MB_BM - 1) / MB_BM, G = gridDim.x;

    // ---- weight ring: mlp_fused.h's, with W1 := fc2.weight^T [H, E] and W2 := fc1.weight^T [E, H]
    const int dr = lane >> 3, dp = lane & 7;
    const int drow = 8 * w + dr;
    const unsigned drow2_ = (unsigned)(2 * drow), swz16_ = (unsigned)((dp ^ mlp_swz(drow)) * 16);
    int slot_i = 0, slot_c = 0, pos_i = 0;
    const char* req_base = nullptr;
    long req_step_a = 0, req_step_b = 0;
    unsigned req_lane = 0;
    char* req_lds = nullptr;
    auto issue_prepare = [&]() __attribute__((always_inline)) {
        int is_p2, chunk, half;
        const unsigned drow2 = (unsigned)opaque_vgpr((int)drow2_), swz16 = (unsigned)opaque_vgpr((int)swz16_);
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
    auto issue_one = [&](int i) __attribute__((always_inline)) {
        glds16(req_base + ((i & 1) * req_step_a + (i >> 1) * req_step_b) + req_lane, req_lds + 4096 * i);
    };
    const unsigned smem_addr = lds_addr_of(smem);
    // `with_u`: the window three back was followed by a second-product window's four u requests (see the header)
    auto acquire = [&](bool with_u) __attribute__((always_inline)) -> unsigned {
        if (with_u) glds_wait<WAIT_RING_U>();
        else glds_wait<WAIT_RING>();
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
    const int prow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);      // bits 2 and 3 of the row index swapped
    const unsigned off1_0 = (unsigned)(prow * 128 + ((hf ^ mlp_swz(prow)) * 16));
    const unsigned off2_0 = (unsigned)(lq * 128 + ((hf ^ mlp_swz(lq)) * 16));
    const float inv_e = 1.0f / (float)E;
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
    const buf_rsrc rs_du = make_rsrc(p.du, (unsigned)((((long)p.M - 1) * p.lddu + p.H) * 2));
    const unsigned lut_addr = lds_addr_of(lut) - 4u * MLP_LUT_LO;       // byte address of entry "magnitude 0"

    // ---- u requests: instruction j of chunk c of the tile whose first row is m0 = this wave's rows 8 j .. + 7, 128 bytes each, into
    // image `buf`; LDS position dp of image row r holds the logical 16-byte slot dp ^ (r & 7) (the slot order LaneOff::scr_wr reads)
    auto u_request = [&](int j, int m0w, int c, int buf) __attribute__((always_inline)) {
        const int ln = opaque_vgpr(t) & 63, rr = ln >> 3, pp = ln & 7;
        int row = m0w + 8 * j + rr;
        row = row < p.M ? row : p.M - 1;
        glds16(reinterpret_cast<const char*>(p.u) + ((long)row * p.ldu + 64 * c) * 2 + ((pp ^ rr) * 16), ubuf + buf * MB_UBUF + j * 1024);
    };

    bool first_tile = true;
    for (int tile = blockIdx.x; tile < tiles; tile += G) {
        const int m0 = tile * MB_BM, r0 = m0 + 32 * w;
        const int row = r0 + lq, grow = row < p.M ? row : p.M - 1;
        int next_r0 = (tile + G < tiles ? tile + G : tile) * MB_BM + 32 * w;      // whose chunks 0 / 1 the last two chunks request
        if (first_tile) {                                                         // nobody requested this tile's first two chunks
#pragma unroll
            for (int j = 0; j < 4; ++j) u_request(j, r0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) u_request(j, r0, NC > 1 ? 1 : 0, 1);
            first_tile = false;
        }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        {
            bf16x8 yf[KJ];                 // this lane's row of gb as B operands of the first product
            {
                const buf_rsrc rs_a = make_rsrc(p.gb, opaque_u32((unsigned)((((long)p.M - 1) * p.ld_gb_in + E) * 2)));
                const unsigned so = (unsigned)r0 * (unsigned)(p.ld_gb_in * 2), lo_y = LaneOff(t).frag(p.ld_gb_in, 2, 8);
#pragma unroll
                for (int j = 0; j < KJ; ++j) yf[j] = __builtin_bit_cast(bf16x8, stream_load16<NT_MLP_Y>(rs_a, lo_y, so + 32 * j));
            }
            f32x16 h[2];
            u32x4 hbw[4];                  // du of the chunk being consumed, as packed bf16 B operands of the second product
            auto p1_piece = [&](auto KH, bool with_u, auto extra, auto filler) {
                constexpr int kh = decltype(KH)::value;
                using Extra = decltype(extra);
                const unsigned sb = acquire(with_u);
                const unsigned a0_ = sb + off1_0, areg[4] = {a0_, a0_ ^ 32u, a0_ ^ 64u, a0_ ^ 96u};
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
            auto p2_piece = [&](auto HH, bool with_u, auto filler) {
                constexpr int hh = decltype(HH)::value;
                const unsigned sb = acquire(with_u);
                const unsigned a0_ = sb + off2_0, areg[4] = {a0_, a0_ ^ 32u, a0_ ^ 64u, a0_ ^ 96u};
                mlp_product<4 * NTH, DEPTH, MlpMapP2<NTH>, MlpNoExtra>(
                    areg,
                    [&](auto K, const bf16x8& a) {
                        constexpr int k = decltype(K)::value;
                        acc[NTH * hh + k % NTH] = mfma_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, hbw[k / NTH]), acc[NTH * hh + k % NTH]);
                    },
                    [&](auto K) {
                        dma_slot(K, std::integral_constant<int, 4 * NTH>{});
                        filler(K);
                    });
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            p1_piece(I0{}, false, MlpNoExtra{}, [](auto) {});
            p1_piece(I1{}, false, MlpNoExtra{}, [](auto) {});
#pragma unroll 1
            for (int c = 0; c < NC; ++c) {
                // element-wise step of chunk c, two elements ("a pair") at a time: registers 8 s + (0 .. 7) of tile tt are hidden
                // units 64 c + 32 tt + 16 s + 8 hf + (0 .. 7); dword e of ucur[2 tt + s] holds u of units 2 e, 2 e + 1 of those 8.
                // A pair is ISSUED (two gathers of gelu'(|u|)) at one MFMA step of the NEXT chunk's first product and FINISHED LAG steps
                // later - mlp_fused.h's schedule.
                const int ub = c & 1;
                u32x4 ucur[4];
                {
                    const LaneOff lo(t);
                    char* img = ubuf + ub * MB_UBUF;
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) ucur[k4] = *reinterpret_cast<const u32x4*>(img + lo.scr_wr(2 * k4 + lo.hf));
                    lds_drain();           // (the product loops count their own LDS operations from an empty queue)
                }
                f32x16 hv[2] = {h[0], h[1]};   // dh; overwritten by du (fp32) pair by pair for the column sums
                float t0[8], t1[8];
                auto pair_issue = [&](auto PI) {
                    constexpr int pi = decltype(PI)::value, k4 = pi >> 2, e = pi & 3, sl = pi & 7;
                    const unsigned uw = ucur[k4][e];
                    unsigned m0_ = uw & 0x7fffu, m1_ = (uw >> 16) & 0x7fffu;
                    m0_ = m0_ < MLP_LUT_LO ? MLP_LUT_LO : (m0_ > MLP_LUT_HI - 1u ? MLP_LUT_HI - 1u : m0_);
                    m1_ = m1_ < MLP_LUT_LO ? MLP_LUT_LO : (m1_ > MLP_LUT_HI - 1u ? MLP_LUT_HI - 1u : m1_);
                    lds_gather_f32(t0[sl], lut_addr + 4u * m0_);
                    lds_gather_f32(t1[sl], lut_addr + 4u * m1_);
                };
                auto pair_finish = [&](auto PI) {
                    constexpr int pi = decltype(PI)::value, k4 = pi >> 2, e = pi & 3, r = 8 * (k4 & 1) + 2 * e, sl = pi & 7;
                    lds_landed(t0[sl], t1[sl]);
                    const unsigned uw = ucur[k4][e];
                    // gelu'(-u) = 1 - gelu'(u): the sign bit of the bf16 pattern selects
                    const float d0 = (uw & 0x8000u) ? 1.0f - t0[sl] : t0[sl], d1 = (uw & 0x80000000u) ? 1.0f - t1[sl] : t1[sl];
                    const float du0 = hv[k4 >> 1][r] * d0, du1 = hv[k4 >> 1][r + 1] * d1;
                    hv[k4 >> 1][r] = du0;
                    hv[k4 >> 1][r + 1] = du1;
                    hbw[k4][e] = pack_bf2(du0, du1);
                    if (e == 3) {          // [32 rows][64 hidden] bf16 image of du where this chunk's u was
                        const LaneOff lo(t);
                        *reinterpret_cast<u32x4*>(ubuf + ub * MB_UBUF + lo.scr_wr(2 * k4 + lo.hf)) = hbw[k4];
                    }
                };
                if (c + 1 < NC) {
                    auto piece_with_gelu = [&](auto KH, bool with_u) {
                        constexpr int kh = decltype(KH)::value;
                        p1_piece(KH, with_u, MlpGeluSchedule<KJ, DEPTH>{}, [&](auto K) {
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
                    piece_with_gelu(I0{}, c > 0);      // W0: behind the previous chunk's second-product window and its u requests
                    piece_with_gelu(I1{}, false);      // W1
                } else {
                    mlp_static_for<0, 4>([&](auto Gq) {
                        constexpr int gq_ = decltype(Gq)::value;
                        mlp_static_for<4 * gq_, 4 * gq_ + 4>(pair_issue);
                        lds_drain();
                        mlp_static_for<4 * gq_, 4 * gq_ + 4>(pair_finish);
                    });
                }
                {   // F: the du image leaves as 128-byte row segments
                    wave_lds_fence();
                    const LaneOff lo(t);
                    const unsigned lo_u = lo.rows8(p.lddu, 2);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(ubuf + ub * MB_UBUF + lo.scr_rd(i));
                        stream_store16<NT_MLP_U>(rs_du, lo_u, (unsigned)(r0 + 8 * i) * (unsigned)(p.lddu * 2) + 128 * c, v);
                    }
                    wave_lds_fence();
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
                // W2 + its four u requests into the image that has just been read out: chunk c + 2 of this tile, or chunk 0 / 1 of the
                // workgroup's next tile (a workgroup's last tile re-reads its own: the requests keep the window's count)
                const int uc = c + 2 < NC ? c + 2 : c + 2 - NC, ur0 = c + 2 < NC ? r0 : next_r0;
                p2_piece(I0{}, false, [&](auto K) {
                    constexpr int k = decltype(K)::value;
                    if constexpr (k % 4 == 3 && k < 16) u_request(k / 4, ur0, uc, ub);
                });
                p2_piece(I1{}, true, [](auto) {});     // W3: behind W2's ten requests
            }
        }
        // ---- epilogue: LayerNorm-2 backward on acc = dy2 (rowgemm.h's RG_LNBWD passes on the bf16 stream; rows are complete inside
        // lanes l, l ^ 32).  Scratch image: the ring slot of the piece consumed last - every wave is done with it behind this barrier and
        // its refill is issued behind the next tile's first barrier (both u images already hold the next tile's first chunks).
        lds_barrier();
        char* scratch = smem + (slot_c == 0 ? NSLOT - 1 : slot_c - 1) * PIECE + w * 4096;
        const float mu = p.mean[grow], rs = p.rstd[grow];
        const buf_rsrc rs_x = make_rsrc(p.x, opaque_u32((unsigned)((((long)p.M - 1) * p.ldx + E) * 4)));
        const buf_rsrc rs_g = make_rsrc(p.g, opaque_u32((unsigned)((((long)p.M - 1) * p.ldg + E) * 2)));
        const buf_rsrc rs_b = make_rsrc(p.gb_out, p.gb_out ? opaque_u32((unsigned)((((long)p.M - 1) * p.ld_gbo + E) * 2)) : 0u);
        const unsigned so_x = (unsigned)r0 * (unsigned)(p.ldx * 4);
        float s1 = 0.f, sq = 0.f;
        {
            const unsigned lo_x = LaneOff(t).frag(p.ldx, 4, 4);
            constexpr int PA = 4;
            u32x4 xb[PA][4];
            auto load_x = [&](int nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) xb[nt % PA][g] = stream_load16<NT_RG_X>(rs_x, lo_x, so_x + (32 * nt + 8 * g) * 4);
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
                    const f32x4v ga = *reinterpret_cast<const f32x4v*>(vga + 32 * nt + 8 * g + 4 * hf);
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
            const unsigned lo_gl = lo.frag(p.ldg, 2, 4), so_g = (unsigned)r0 * (unsigned)(p.ldg * 2);
            const unsigned lo_o = lo.rows8(p.ldg, 2), lo_n = lo.rows8(p.ld_gbo, 2);
            constexpr int PB = 3;
            buf_u32x2 gbuf[PB][4];
            auto load_g = [&](int nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    gbuf[nt % PB][g] = buf_u32x2{0u, 0u};
                    if (p.accumulate) gbuf[nt % PB][g] = buf_load8(rs_g, lo_gl, so_g + (32 * nt + 8 * g) * 2);
                }
            };
#pragma unroll
            for (int nt = 0; nt < PB - 1 && nt < NT; ++nt) load_g(nt);
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                u32x2 ypk[2][4], gpk[2][4];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int nt = 2 * np + tt;
                    if (nt + PB - 1 < NT) load_g(nt + PB - 1);
                    float vbi[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float oo[4] = {bf_lo(gbuf[nt % PB][g].x), bf_hi(gbuf[nt % PB][g].x), bf_lo(gbuf[nt % PB][g].y), bf_hi(gbuf[nt % PB][g].y)};
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float pf = acc[nt][4 * g + e];
                            const unsigned pk = __builtin_bit_cast(unsigned, pf);
                            const float dx = fmaf(-c2, rg_unpack_xh(pk), fmaf(rg_unpack_dg(pk), rs, -c1));
                            v[e] = oo[e] + dx;
                        }
                        gpk[tt][g].x = pack_bf2(v[0], v[1]);
                        gpk[tt][g].y = pack_bf2(v[2], v[3]);
                        if (p.gb_out) {
                            ypk[tt][g].x = pack_bf2(v[0] * sc, v[1] * sc);
                            ypk[tt][g].y = pack_bf2(v[2] * sc, v[3] * sc);
                            vbi[4 * g] = bf_lo(ypk[tt][g].x); vbi[4 * g + 1] = bf_hi(ypk[tt][g].x);
                            vbi[4 * g + 2] = bf_lo(ypk[tt][g].y); vbi[4 * g + 3] = bf_hi(ypk[tt][g].y);
                        }
                    }
                    if (p.gb_out && p.dbias) rg_colsum16(vbi, cs + 2 * E + 32 * nt, lq, hf);
                    CCD_SCHED_FENCE();
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<u32x2*>(scratch + lo.scr_wr(4 * tt + g) + 8 * lo.hf) = gpk[tt][g];
                wave_lds_fence();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                    stream_store16<NT_RG_G>(rs_g, lo_o, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldg * 2) + 128 * np, o);
                }
                wave_lds_fence();
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
                        stream_store16<NT_RG_GB>(rs_b, lo_n, (unsigned)(r0 + 8 * i) * (unsigned)(p.ld_gbo * 2) + 128 * np, o);
                    }
                    wave_lds_fence();
                }
            }
        }
    }
    glds_wait_all();                       // requested pieces / u images that no tile consumed must not outlive the workgroup's LDS
    __syncthreads();
    for (int i = t; i < E; i += MB_THREADS) {
        atomicAdd(p.dgamma + i, cs[i]);
        atomicAdd(p.dbeta + i, cs[E + i]);
        if (p.dbias) atomicAdd(p.dbias + i, cs[2 * E + i]);
    }
    // every workgroup starts its sweep at another column (256 workgroups adding to the same address at the same moment serialise in the L2)
    const int rot = (int)(((unsigned)blockIdx.x * 2654435761u >> 8) % (unsigned)p.H);
    for (int i = t; i < p.H; i += MB_THREADS) {
        int c = i + rot;
        c = c >= p.H ? c - p.H : c;
        const float v = csh[c];
        if (v != 0.f) atomicAdd(p.db1 + c, v);
    }
}

}  // namespace ccd
