// rowproj.h - the K = E projections of a transformer block with their ACTIVATION rows resident in registers:
//     out[M, N] (bf16) = a[M, E] . W[N, E]^T + bias          (qkv = LayerNorm-1 output . Wqkv^T, vision_transformer.py:84;
//                                                            d_att = gb . Wproj, the data gradient of :91)
// Round 2 ran these on the 256 x 256 tile of gemm256.h: with K = 384 a tile is 6 k-steps long, so its DMA ramp, the epilogue
// and the barriers weigh as much as the products (0.16 ms for the qkv shape, 725 TFLOP/s, and the A panel is re-read by every
// column tile).  Here the structure of the fused MLP's first product (mlp_fused.h) is used on its own:
//   * "row owners": one workgroup = 4 waves, one wave per SIMD; a wave keeps RB blocks of 32 rows of `a` in registers as MFMA
//     B operands for the whole tile (RB * E / 4 VGPRs) - `a` is read from HBM exactly once, no LDS staging, no k-loop ramp;
//   * only the WEIGHTS move: pieces of 64 output columns x one K half (32 * E * 2 bytes) stream through the 5-slot LDS ring by
//     LDS-DMA (counted vmcnt, one LDS-only barrier per piece), shared by the four waves; the product is computed transposed
//     (OUT^T[64 cols][32 rows] = W[cols] . a^T) so a lane owns a row and its accumulators hold 8 consecutive columns;
//   * RB = 2: every weight fragment read from LDS feeds TWO MFMAs (one per row block).  In the fused MLP (RB = 1) four waves
//     reading a 1-KiB fragment per 32-cycle MFMA ask the LDS for exactly its 128 B / clk peak; here they ask for half;
//   * a finished 64-column chunk leaves through the wave's 4-KiB scratch image as 128-byte row segments (8 rows per store
//     instruction) - ONE CHUNK LATE (round 4): at its end a chunk is only packed to bf16 (32 registers); its scratch writes, scratch
//     reads and stores are issued between the MFMAs of the NEXT chunk's two pieces (row block rb under piece rb), on the product's own
//     counted LDS queue.  Issued in a burst at the chunk's end, the 8 stores of all 256 CUs (8 MB) arrived at HBM together: the
//     product phases left HBM idle and the store phases left the MFMAs idle (0.143 ms for the qkv shape, 0.093 with the stores
//     removed); spread over the next product they cost nothing.
// Ring protocol, fragment scheduling (mlp_product: reads issued 6 steps ahead, counted lgkmcnt) and the row-order trick that
// makes 8 consecutive columns land in consecutive accumulator registers (bits 2 and 3 of the weight row index swapped) are
// mlp_fused.h's.
#pragma once

namespace ccd {

struct RowProjParams {
    const bf16_t* a;        // [M, E] bf16
    long lda;
    const bf16_t* w;        // [N, E] bf16 (nn.Linear weight, or the transposed mirror for a data gradient)
    long ldw;
    const float* bias;      // [N] or null
    bf16_t* out;            // [M, N] bf16
    long ldc;
    int M, N;
};

constexpr int RP_THREADS = 256, RP_SCRATCH = 4096;
__host__ __device__ constexpr int rp_rows(int RB) { return 4 * 32 * RB; }              // rows of a workgroup tile
__host__ __device__ inline int rp_smem_bytes(int E, int N) { return mlp_slots(E) * mlp_piece_bytes(E) + 4 * RP_SCRATCH + N * 4; }

// the late chunk's LDS operations inside a piece of KJ >= 24 steps: scratch writes behind steps 2, 3, 4, 6, scratch reads behind 13 .. 16
// (every write has retired when the wait of step 13 passes: read 13 was issued behind the last write), stores behind 20 .. 23 (the
// read issued behind step 13 + i has retired at the wait of step 20 + i)
struct RpLateExtra {
    static constexpr int at(int k) { return (k == 2 || k == 3 || k == 4 || k == 6 || (k >= 13 && k <= 16)) ? 1 : 0; }
    static constexpr int write_unit(int k) { return k == 2 ? 0 : k == 3 ? 1 : k == 4 ? 2 : k == 6 ? 3 : -1; }
};

template <int E, int RB>
__global__ __launch_bounds__(RP_THREADS, 1) void rowproj_kernel(RowProjParams p) {
    constexpr int KT = E / 64;             // DMA instructions per wave and piece
    constexpr int KJ = E / 16;             // MFMA k-steps over K; a piece (one K half, 2 column tiles) is KJ fragments
    constexpr int PIECE = mlp_piece_bytes(E);
    constexpr int NSLOT = mlp_slots(E);
    constexpr int AHEAD = NSLOT - 1;
    constexpr int DEPTH = 6;
    constexpr int BM = rp_rows(RB);
    static_assert(E % 128 == 0 && AHEAD >= 2 && (RB == 1 || RB == 2), "ring bookkeeping");
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, hf = lane >> 5, lq = lane & 31;
    const int w = uniform_i32(t >> 6);
    char* scratch = smem + NSLOT * PIECE + w * RP_SCRATCH;
    float* vb = reinterpret_cast<float*>(smem + NSLOT * PIECE + 4 * RP_SCRATCH);
    for (int i = t; i < p.N; i += RP_THREADS) vb[i] = p.bias ? p.bias[i] : 0.f;
    __syncthreads();

    const int NC = p.N / 64, NP = 2 * NC;  // 64-column chunks, pieces per row tile
    const int tiles = (p.M + BM - 1) / BM, G = gridDim.x;

    // ---- DMA (as mlp_fused.h's W1 pieces): piece = [KT / 2 k-tiles of one K half][64 weight rows][128 B]; wave w moves the
    // instructions whose 8-row block index is w modulo 4: instruction i = rows 32 (i & 1) + 8 w .. + 7 of k-tile i >> 1
    const int dr = lane >> 3, dp = lane & 7;
    const int drow = 8 * w + dr;
    const unsigned req_lane = (unsigned)(2 * drow) * (unsigned)p.ldw + (unsigned)((dp ^ mlp_swz(drow)) * 16);
    int slot_i = 0, slot_c = 0, pos_i = 0;
    const char* req_base = nullptr;
    char* req_lds = nullptr;
    const long step_a = 64 * p.ldw;        // bytes between the two 32-row halves of a chunk: 32 rows * ldw * 2
    auto issue_prepare = [&]() {
        const int chunk = pos_i >> 1, half = pos_i & 1;
        req_base = reinterpret_cast<const char*>(p.w) + ((long)(64 * chunk) * p.ldw + half * (E / 2)) * 2;
        req_lds = smem + slot_i * PIECE + w * 1024;
        slot_i = slot_i + 1 == NSLOT ? 0 : slot_i + 1;
        pos_i = pos_i + 1 == NP ? 0 : pos_i + 1;
    };
    auto issue_one = [&](int i) { glds16(req_base + ((i & 1) * step_a + (i >> 1) * 128) + req_lane, req_lds + 4096 * i); };
    const unsigned smem_addr = lds_addr_of(smem);
    auto acquire = [&]() -> unsigned {
        glds_wait<(AHEAD - 1) * KT>();     // my quarter of the next piece has landed (stores in the window only make this stricter)
        lds_barrier();                     // everybody's has, and everybody is done with the previous piece
        issue_prepare();                   // whose slot is re-filled AHEAD pieces ahead, between the MFMAs below
        const unsigned sb = smem_addr + (unsigned)(slot_c * PIECE);
        slot_c = slot_c + 1 == NSLOT ? 0 : slot_c + 1;
        return sb;
    };
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
        issue_prepare();
#pragma unroll
        for (int i = 0; i < KT; ++i) issue_one(i);
    }
    // fragment read offsets: weight row lq of a 32-row tile is fed in the order that swaps bits 2 and 3 of the row index, so
    // that accumulator registers 8 s + (0 .. 7) of tile tt hold columns 32 tt + 16 s + 8 hf + (0 .. 7) of the chunk
    const int prow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    unsigned off1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) off1[kk] = (unsigned)(prow * 128 + (((2 * kk + hf) ^ mlp_swz(prow)) * 16));
    const buf_rsrc rs_a = make_rsrc(p.a, (unsigned)((((long)p.M - 1) * p.lda + E) * 2));
    const buf_rsrc rs_o = make_rsrc(p.out, (unsigned)((((long)p.M - 1) * p.ldc + p.N) * 2));

    // the chunk that has not left yet: packed bf16 (register 4 * tt + 2 * s + (0, 1) pairs of columns, as the scratch writes want them),
    // the byte offset of its first row and column (BUF_OOB: none)
    constexpr bool LATE = KJ >= 24;
    u32x4 pend[RB][4];
    unsigned pend_so[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        pend_so[rb] = BUF_OOB;
#pragma unroll
        for (int j = 0; j < 4; ++j) pend[rb][j] = u32x4{0u, 0u, 0u, 0u};
    }
    const unsigned scr_addr = lds_addr_of(scratch);
    for (int tile = blockIdx.x; tile < tiles; tile += G) {
        const int r0 = tile * BM + 32 * RB * w;              // this wave's first row
        bf16x8 af[RB][KJ];                                   // row lq of block rb: k = 16 j + 8 hf .. + 7
        {
            const unsigned lo_a = (unsigned)((lq * p.lda + 8 * hf) * 2);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const unsigned so = (unsigned)(r0 + 32 * rb) * (unsigned)(p.lda * 2);
#pragma unroll
                for (int j = 0; j < KJ; ++j) af[rb][j] = __builtin_bit_cast(bf16x8, stream_load16<NT_RP_A>(rs_a, lo_a, so + 32 * j));
            }
        }
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            f32x16 h[RB][2];
            {   // accumulators start at the bias: register r of tile tt is column 64 c + 32 tt + 16 (r >> 3) + 8 hf + (r & 7)
                const float* bp = vb + 64 * c + 8 * hf;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4v b = *reinterpret_cast<const f32x4v*>(bp + 32 * tt + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) {
                            h[rb][tt][4 * q] = b.x; h[rb][tt][4 * q + 1] = b.y; h[rb][tt][4 * q + 2] = b.z; h[rb][tt][4 * q + 3] = b.w;
                        }
                    }
            }
            auto piece = [&](auto KH) {
                constexpr int kh = decltype(KH)::value;
                const unsigned sb = acquire();
                const unsigned areg[4] = {sb + off1[0], sb + off1[1], sb + off1[2], sb + off1[3]};
                if constexpr (LATE && kh < RB) {
                    // row block kh of the PREVIOUS chunk leaves under this piece
                    const int ln = opaque_vgpr(t) & 63, lhf = ln >> 5, llq = ln & 31, ldr_ = ln >> 3, ldp = ln & 7;
                    const unsigned wr_base = scr_addr + (unsigned)(llq * 128), rd_base = scr_addr + (unsigned)(ldr_ * 128 + ((ldp ^ ldr_) * 16));
#if defined(CCD_RP_LAB) && (CCD_RP_LAB & 4)      // lab build: every (tile, chunk, wave, row block) leaves as ONE contiguous 4 KiB (not the matrix layout)
                    const unsigned lo_o = (unsigned)(ldr_ * 128 + ldp * 16);
                    const unsigned row_step = 8 * 128;
#else
                    const unsigned lo_o = (unsigned)(ldr_ * p.ldc * 2 + ldp * 16);
                    const unsigned row_step = 8u * (unsigned)(p.ldc * 2);
#endif
                    u32x4 ob[4];
                    mlp_product<KJ, DEPTH, MlpMapP1<KT / 2>, RpLateExtra>(
                        areg,
                        [&](auto K, const bf16x8& a) {
                            constexpr int k = decltype(K)::value;
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb)
                                h[rb][k & 1] = mfma_32x32x16_bf16(a, af[rb][(KJ / 2) * kh + (k >> 1)], h[rb][k & 1]);
                        },
                        [&](auto K) {
                            constexpr int k = decltype(K)::value, stride = KJ / KT;
                            if constexpr (k % stride == 1 && k / stride < KT) issue_one(k / stride);
                            if constexpr (RpLateExtra::write_unit(k) >= 0) {
                                constexpr int j = RpLateExtra::write_unit(k);           // unit (tt, s) = (j >> 1, j & 1): columns 8 * slot16 .. + 7
                                lds_write16(wr_base + (unsigned)((((2 * j + lhf) ^ (llq & 7))) * 16), pend[kh][j]);
                            }
                            if constexpr (k >= 13 && k <= 16) lds_read16(ob[k - 13], rd_base + (unsigned)((k - 13) * 8 * 128));
                            if constexpr (k >= 20 && k <= 23) {
                                needed_here(ob[k - 20]);
                                stream_store16<NT_RP_OUT>(rs_o, lo_o, pend_so[kh] + (unsigned)(k - 20) * row_step, ob[k - 20]);
                            }
                        });
                } else
                mlp_product<KJ, DEPTH, MlpMapP1<KT / 2>, MlpNoExtra>(
                    areg,
                    [&](auto K, const bf16x8& a) {
                        constexpr int k = decltype(K)::value;
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb)
                            h[rb][k & 1] = mfma_32x32x16_bf16(a, af[rb][(KJ / 2) * kh + (k >> 1)], h[rb][k & 1]);
                    },
                    [&](auto K) {
                        constexpr int k = decltype(K)::value, stride = KJ / KT;
                        if constexpr (k % stride == 1 && k / stride < KT) issue_one(k / stride);
                    });
            };
            piece(std::integral_constant<int, 0>{});
            piece(std::integral_constant<int, 1>{});
            if constexpr (LATE) {
                // ---- the chunk is packed and waits for the next chunk's products (a row block that RB < 2 pieces cannot carry would leave here)
                static_assert(RB <= 2, "one row block per piece of the next chunk");
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int sq = 0; sq < 2; ++sq)
#pragma unroll
                            for (int e = 0; e < 4; ++e) pend[rb][2 * tt + sq][e] = pack_bf2(h[rb][tt][8 * sq + 2 * e], h[rb][tt][8 * sq + 2 * e + 1]);
                    pend_so[rb] = (unsigned)(r0 + 32 * rb) * (unsigned)(p.ldc * 2) + 128 * c;
#ifdef CCD_RP_LAB       // lab build: the stores are issued but fall outside the descriptor (no HBM writes) / contiguous blocks
                    if (CCD_RP_LAB & 2) pend_so[rb] = BUF_OOB;
                    if (CCD_RP_LAB & 4) pend_so[rb] = (unsigned)((((tile * NC + c) * 4 + w) * RB + rb) * 4096);
#endif
                }
                continue;
            }
            // ---- the chunk leaves: [32 rows][64 columns] bf16 image in the wave's scratch -> 128-byte row segments
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int ln = opaque_vgpr(t) & 63, lhf = ln >> 5, llq = ln & 31, ldr_ = ln >> 3, ldp = ln & 7;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        u32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = pack_bf2(h[rb][tt][8 * s + 2 * e], h[rb][tt][8 * s + 2 * e + 1]);
                        const int slot16 = 4 * tt + 2 * s + lhf;             // columns 8 * slot16 .. + 7
                        *reinterpret_cast<u32x4*>(scratch + llq * 128 + ((slot16 ^ (llq & 7)) * 16)) = v;
                    }
                wave_lds_fence();
                const unsigned lo_o = (unsigned)(ldr_ * p.ldc * 2 + ldp * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(scratch + (ldr_ + 8 * i) * 128 + ((ldp ^ ldr_) * 16));
                    stream_store16<NT_RP_OUT>(rs_o, lo_o, (unsigned)(r0 + 32 * rb + 8 * i) * (unsigned)(p.ldc * 2) + 128 * c, v);
                }
                wave_lds_fence();
            }
        }
    }
    if constexpr (LATE) {                  // the last chunk of the workgroup's last tile: nothing left to hide it under
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int ln = opaque_vgpr(t) & 63, lhf = ln >> 5, llq = ln & 31, ldr_ = ln >> 3, ldp = ln & 7;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<u32x4*>(scratch + llq * 128 + (((2 * j + lhf) ^ (llq & 7)) * 16)) = pend[rb][j];
            wave_lds_fence();
            const unsigned lo_o = (unsigned)(ldr_ * p.ldc * 2 + ldp * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(scratch + (ldr_ + 8 * i) * 128 + ((ldp ^ ldr_) * 16));
                stream_store16<NT_RP_OUT>(rs_o, lo_o, pend_so[rb] + (unsigned)(8 * i) * (unsigned)(p.ldc * 2), v);
            }
            wave_lds_fence();
        }
    }
    glds_wait_all();                       // requested pieces that no tile consumed must not outlive the workgroup's LDS
}

}  // namespace ccd
