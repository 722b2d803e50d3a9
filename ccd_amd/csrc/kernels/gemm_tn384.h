// gemm_tn384.h - the weight-gradient product of the ViT blocks:  C[P,Q] += alpha * sum_m A[m,P]^T . B[m,Q]   (fp32 atomics)
// for P % 384 == 0, Q % 192 == 0, contraction length % 32 == 0 - dW = dY^T X of qkv (1152 x 384), proj (384 x 384),
// fc1 (1536 x 384) and fc2 (384 x 1536) at embed_dim 384 (Dino/modules/svtr.py:92-145 run backwards by autograd).
//
// Why a second TN kernel next to gemm.h's 128-square one: the kind is bound by HBM - PQ / (P + Q)
// = 192 ... 307 flop per byte, the machine balance is ~310 - and the 128-square kernel fetched 1.7 - 1.9 x its algorithmic bytes
// (27 - 36 tiles per contraction slice sharing rows through an L2 that most slices straddled) with two k-tiles in flight per
// workgroup.  Here
//   * an output tile is 384 x 192 (8 waves as 4 x 2, 96 x 96 = 3 x 3 MFMA tiles per wave): 2 / 6 / 8 / 8 tiles cover the whole
//     output, so ONE group of that many workgroups reads every dY / X row of its contraction slice exactly once between them,
//   * a group lives on ONE XCD (workgroup b runs on XCD b % 8): its members start together, stream the same rows at the same rate
//     and meet in that XCD's L2 - a member that falls behind hits what the others fetched, one that runs ahead misses and waits,
//   * operands go HBM -> LDS by DMA (global_load_lds, 16 B per lane) exactly as they lie in memory - [32 contraction rows][columns]
//     - four 36-KiB stages, three in flight (108 KiB per CU), counted vmcnt, one LDS barrier per stage,
//   * the MFMA fragments (k along the registers) come out of that k-major image through ds_read_b64_tr_b16: in a 16-lane group
//     lane j addresses the 4 bf16 (row j / 4, columns 4 (j % 4) ..), lane i receives column i of the four rows (tools/probe/
//     tr_probe.hip), two reads per fragment - no register transpose, no ds_write pass, no staging VGPRs.
// LDS image of a stage: A part 32 rows x 768 B, B part 32 rows x 384 B.  The four rows of one transposing read lie 768 B (= 0 mod
// 256) resp. 384 B (= 128 mod 256) apart, i.e. on the same / on two alternating halves of the 64 banks: the 16-byte chunk index is
// XORed with (row & 3) << 2 (A) resp. ((row >> 1) & 1) << 2 (B) - applied to the DMA's SOURCE address, the DMA itself writes
// lane-linearly - which puts the 4 rows x 64 B of a half-wave on 4 different 64-byte bank groups.
#pragma once

namespace ccd {

constexpr int TN3_TP = 384, TN3_TQ = 192, TN3_BK = 32, TN3_THREADS = 512, TN3_STAGES = 4;
constexpr int TN3_A_ROWB = TN3_TP * 2, TN3_B_ROWB = TN3_TQ * 2;
constexpr int TN3_A_BYTES = TN3_BK * TN3_A_ROWB, TN3_B_BYTES = TN3_BK * TN3_B_ROWB;        // 24 KiB + 12 KiB
constexpr int TN3_STAGE_BYTES = TN3_A_BYTES + TN3_B_BYTES;                                 // 36 KiB
constexpr int TN3_SMEM_BYTES = TN3_STAGES * TN3_STAGE_BYTES;                               // 144 KiB
constexpr int TN3_A_PIECES = TN3_A_BYTES / 1024 / 8;                                       // 1-KiB DMA pieces per wave and stage: 3
constexpr int TN3_PER_STAGE = TN3_A_PIECES + 2;                                            // + 2 B pieces (waves 4-7: one)

__device__ __forceinline__ int tn3_swz_a(int row) { return (row & 3) << 2; }
__device__ __forceinline__ int tn3_swz_b(int row) { return ((row >> 1) & 1) << 2; }

// p.M = P, p.N = Q, p.K = contraction length, p.k_per_split = rows per slice (multiple of 32), p.work_items = number of slices,
// p.m_fastest = XCDs the grid is spread over (8, or 1: every workgroup is its own group member in launch order).
// p.M2 > 0: a second problem (A2, B2, C2; same K) whose tiles follow the first one's inside every group - the launch then pays
// ONE atomic epilogue of 288 KiB per workgroup for two products (the epilogue, ~55 us, is a third of a single product's time).
// Grid: xcds * slots; workgroup b -> xcd b % xcds, slot b / xcds; slot -> (group, tile); slice = xcd * groups_per_xcd + group.
__global__ __launch_bounds__(TN3_THREADS, 1) void gemm_tn384_kernel(GemmParams p) {
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hf = lane >> 5, lq = lane & 31;
    const int wm = w & 3, wn = w >> 2;
    const int tiles1 = (p.M / TN3_TP) * (p.N / TN3_TQ), tiles = tiles1 + (p.M2 / TN3_TP) * (p.N2 / TN3_TQ);
    const int xcds = p.m_fastest, spx = (int)gridDim.x / xcds, gpx = spx / tiles;
    const int xcd = (int)blockIdx.x % xcds, slot = (int)blockIdx.x / xcds;
    const int group = slot / tiles;
    int tile = slot % tiles;
    const int slice = xcd * gpx + group;
    if (group >= gpx || slice >= p.work_items) return;
    if (tile >= tiles1) {                                    // a tile of the second problem (wave-uniform)
        tile -= tiles1;
        p.A = p.A2; p.B = p.B2; p.lda = p.lda2; p.ldb = p.ldb2; p.N = p.N2; p.C = p.C2; p.ldc = p.ldc2;
    }
    const int tiles_q = p.N / TN3_TQ;
    const int p0 = (tile / tiles_q) * TN3_TP, q0 = (tile % tiles_q) * TN3_TQ;
    const int k_begin = slice * p.k_per_split;
    const int k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
    const int nk = k_end > k_begin ? (k_end - k_begin) / TN3_BK : 0;
    if (nk == 0) return;

    // ---- DMA sources: piece n of a part covers LDS bytes [1024 n, 1024 n + 1024) of that part, lane L its 16-byte chunk L
    const bf16_t* ga[TN3_A_PIECES];
    const bf16_t* gb[2];
#pragma unroll
    for (int i = 0; i < TN3_A_PIECES; ++i) {
        const int byte = 1024 * (TN3_A_PIECES * w + i) + 16 * lane;
        const int row = byte / TN3_A_ROWB, pos = (byte % TN3_A_ROWB) >> 4;
        ga[i] = p.A + (long)(k_begin + row) * p.lda + p0 + (pos ^ tn3_swz_a(row)) * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int byte = 1024 * (w + 8 * i) + 16 * lane;                 // pieces 0-7: every wave, pieces 8-11: waves 0-3
        const int row = (byte / TN3_B_ROWB) & (TN3_BK - 1), pos = (byte % TN3_B_ROWB) >> 4;
        gb[i] = p.B + (long)(k_begin + row) * p.ldb + q0 + (pos ^ tn3_swz_b(row)) * 8;
    }
    const long a_step = (long)TN3_BK * p.lda, b_step = (long)TN3_BK * p.ldb;
    auto dma = [&](int s) {                                  // requests stage s (called for s = 0, 1, 2, ...: the sources advance)
        char* base = smem + (s & 3) * TN3_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < TN3_A_PIECES; ++i) {
            glds16(ga[i], base + (TN3_A_PIECES * w + i) * 1024);
            ga[i] += a_step;
        }
        glds16(gb[0], base + TN3_A_BYTES + w * 1024);
        gb[0] += b_step;
        if (w < 4) {
            glds16(gb[1], base + TN3_A_BYTES + (8 + w) * 1024);
            gb[1] += b_step;
        }
    };
    // Counted waits see ONLY LDS-DMA operations (5 per stage in waves 0-3, 4 in waves 4-7).  Loads that return to VGPRs must not
    // be mixed into the window: measured here - with 1-byte L2-prefetch loads (and out-of-range buffer loads as padding) between
    // the DMA pieces, `vmcnt(N)` was satisfied by the fast VGPR returns while older DMA pieces were still in flight (NaNs).
    auto wait_steps = [&](int in_flight) {                   // at most `in_flight` stages of this wave's DMA still outstanding
        if (w < 4) {
            if (in_flight >= 2) glds_wait<2 * TN3_PER_STAGE>();
            else if (in_flight == 1) glds_wait<TN3_PER_STAGE>();
            else glds_wait_all();
        } else {
            if (in_flight >= 2) glds_wait<2 * (TN3_PER_STAGE - 1)>();
            else if (in_flight == 1) glds_wait<TN3_PER_STAGE - 1>();
            else glds_wait_all();
        }
    };

    // ---- fragment addresses (see the header): lane (j = lane & 15, half-row group g16, hf)
    const int j = lane & 15, g16 = (lane >> 4) & 1, r4 = j >> 2, c4 = j & 3;
    unsigned base_a[3], base_b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = 8 * hf + r4;
        const int chunk_a = 12 * wm + 4 * i + 2 * g16 + (c4 >> 1);
        base_a[i] = (unsigned)(row * TN3_A_ROWB + ((chunk_a ^ tn3_swz_a(row)) << 4) + (c4 & 1) * 8);
        const int chunk_b = 12 * wn + 4 * i + 2 * g16 + (c4 >> 1);
        base_b[i] = (unsigned)(TN3_A_BYTES + row * TN3_B_ROWB + ((chunk_b ^ tn3_swz_b(row)) << 4) + (c4 & 1) * 8);
    }

    const unsigned smem_addr = lds_addr_of(smem);
    f32x16 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

#ifdef CCD_MLP_LAB    // per-phase cycle totals of waves 0 and 7 (lab bit 4): -> p.colsum_a as 8 u64 per (workgroup < 32, wave)
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define TN3_STAMP(i) if (p.rps_shift & 4) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tprev; tprev = tn_; }
#else
#define TN3_STAMP(i)
#endif
    dma(0);
    if (nk > 1) dma(1);
    if (nk > 2) dma(2);
    wait_steps(nk > 2 ? 2 : (nk > 1 ? 1 : 0));
    lds_barrier();
    for (int kt = 0; kt < ((p.rps_shift & 2) ? 1 : nk); ++kt) {      // (lab bit 2: one stage only)
        if (kt + 3 < nk) dma(kt + 3);                        // into the buffer read in step kt - 1 (every wave passed its barrier)
        TN3_STAMP(0)
        // Fragment reads by hand (prelude: lds_read_tr), 4 per group, LDS returns in order: group order a0 b0 | b1 b2 | a1 a2 of
        // k-step 0, then the same of k-step 1 issued between the MFMA rows of k-step 0; every wait names the youngest fragment
        // its MFMAs need and the number of reads issued behind it.
        const unsigned sb = smem_addr + (unsigned)((kt & 3) * TN3_STAGE_BYTES);
        const unsigned aa0 = sb + base_a[0], aa1 = sb + base_a[1], aa2 = sb + base_a[2];
        const unsigned ab0 = sb + base_b[0], ab1 = sb + base_b[1], ab2 = sb + base_b[2];
        tr_u32x2 x[2][6][2];                                  // [k-step][a0 b0 b1 b2 a1 a2][rows 0-3 / 4-7]
#define TN3_RD_A(ks, slot, addr)                                                     \
        lds_read_tr<(16 * ks) * TN3_A_ROWB>(x[ks][slot][0], addr);                       \
        lds_read_tr<(16 * ks + 4) * TN3_A_ROWB>(x[ks][slot][1], addr);
#define TN3_RD_B(ks, slot, addr)                                                     \
        lds_read_tr<(16 * ks) * TN3_B_ROWB>(x[ks][slot][0], addr);                       \
        lds_read_tr<(16 * ks + 4) * TN3_B_ROWB>(x[ks][slot][1], addr);
#define TN3_FRAG(ks, slot) frag_from_tr(x[ks][slot][0], x[ks][slot][1])
        TN3_RD_A(0, 0, aa0) TN3_RD_B(0, 1, ab0) TN3_RD_B(0, 2, ab1) TN3_RD_B(0, 3, ab2) TN3_RD_A(0, 4, aa1) TN3_RD_A(0, 5, aa2)
        bf16x8 a0 = TN3_FRAG(0, 0), b0 = TN3_FRAG(0, 1), b1 = TN3_FRAG(0, 2), b2 = TN3_FRAG(0, 3), a1 = TN3_FRAG(0, 4), a2 = TN3_FRAG(0, 5);
        lds_wait_frag<8>(b0);
        TN3_STAMP(1)
        acc[0][0] = mfma_32x32x16_bf16(a0, b0, acc[0][0]);
        lds_wait_frag<6>(b1);
        acc[0][1] = mfma_32x32x16_bf16(a0, b1, acc[0][1]);
        lds_wait_frag<4>(b2);
        acc[0][2] = mfma_32x32x16_bf16(a0, b2, acc[0][2]);
        TN3_RD_A(1, 0, aa0) TN3_RD_B(1, 1, ab0)
        lds_wait_frag<6>(a1);
        acc[1][0] = mfma_32x32x16_bf16(a1, b0, acc[1][0]);
        acc[1][1] = mfma_32x32x16_bf16(a1, b1, acc[1][1]);
        acc[1][2] = mfma_32x32x16_bf16(a1, b2, acc[1][2]);
        TN3_RD_B(1, 2, ab1) TN3_RD_B(1, 3, ab2)
        lds_wait_frag<8>(a2);
        acc[2][0] = mfma_32x32x16_bf16(a2, b0, acc[2][0]);
        acc[2][1] = mfma_32x32x16_bf16(a2, b1, acc[2][1]);
        acc[2][2] = mfma_32x32x16_bf16(a2, b2, acc[2][2]);
        TN3_RD_A(1, 4, aa1) TN3_RD_A(1, 5, aa2)
        a0 = TN3_FRAG(1, 0); b0 = TN3_FRAG(1, 1); b1 = TN3_FRAG(1, 2); b2 = TN3_FRAG(1, 3); a1 = TN3_FRAG(1, 4); a2 = TN3_FRAG(1, 5);
        lds_wait_frag<8>(b0);
        acc[0][0] = mfma_32x32x16_bf16(a0, b0, acc[0][0]);
        lds_wait_frag<6>(b1);
        acc[0][1] = mfma_32x32x16_bf16(a0, b1, acc[0][1]);
        lds_wait_frag<4>(b2);
        acc[0][2] = mfma_32x32x16_bf16(a0, b2, acc[0][2]);
        lds_wait_frag<2>(a1);
        acc[1][0] = mfma_32x32x16_bf16(a1, b0, acc[1][0]);
        acc[1][1] = mfma_32x32x16_bf16(a1, b1, acc[1][1]);
        acc[1][2] = mfma_32x32x16_bf16(a1, b2, acc[1][2]);
        lds_wait_frag<0>(a2);
        acc[2][0] = mfma_32x32x16_bf16(a2, b0, acc[2][0]);
        acc[2][1] = mfma_32x32x16_bf16(a2, b1, acc[2][1]);
        acc[2][2] = mfma_32x32x16_bf16(a2, b2, acc[2][2]);
#undef TN3_RD_A
#undef TN3_RD_B
#undef TN3_FRAG
        TN3_STAMP(2)
        const int rem = nk - 1 - kt;                         // stage kt + 1 must have landed before the barrier publishes it
        if (rem > 0) wait_steps((rem < 3 ? rem : 3) - 1);
        TN3_STAMP(3)
        lds_barrier();
        TN3_STAMP(4)
    }
#ifdef CCD_MLP_LAB
    if ((p.rps_shift & 4) && p.colsum_a && blockIdx.x < 32 && lane == 0 && (w == 0 || w == 7)) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.colsum_a) + (blockIdx.x * 2 + (w ? 1 : 0)) * 6;
        for (int i = 0; i < 6; ++i) o[i] = ph[i];
    }
#endif
    glds_wait_all();

    // ---- epilogue: D[p][q], a lane owns column q = lq of 16 rows per tile; 32 lanes = 128 contiguous bytes per atomic
    float* C = reinterpret_cast<float*>(p.C);
    if (p.rps_shift & 1) return;                             // (lab bit 1: no epilogue)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            float* cp = C + (long)(p0 + 96 * wm + 32 * i + 4 * hf) * p.ldc + (q0 + 96 * wn + 32 * jj + lq);
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(cp + (long)((r & 3) + 8 * (r >> 2)) * p.ldc, acc[i][jj][r] * p.alpha);
        }
}

}  // namespace ccd
