// gemm_tn384.h - the weight-gradient product of the ViT blocks:  C[P,Q] += alpha * sum_m A[m,P]^T . B[m,Q]   (fp32 atomics)
// for P % 384 == 0, Q % 192 == 0, contraction length % 32 == 0 - dW = dY^T X of qkv (1152 x 384), proj (384 x 384),
// fc1 (1536 x 384) and fc2 (384 x 1536) at embed_dim 384 (Dino/modules/svtr.py:92-145 run backwards by autograd).
//
// Why a second TN kernel next to gemm.h's 128-square one (docs/LAB_NOTEBOOK.md section 8 item 2): the kind is bound by HBM - PQ / (P + Q)
// = 192 ... 307 flop per byte, the machine balance is ~310 - and the 128-square kernel fetched 1.7 - 1.9 x its algorithmic bytes
// (27 - 36 tiles per contraction slice sharing rows through an L2 that most slices straddled) with two k-tiles in flight per
// workgroup.  Here
//   * an output tile is 384 x 192 (8 waves as 4 x 2, 96 x 96 = 3 x 3 MFMA tiles per wave; or 192 x 192 with 4 waves, two
//     workgroups per CU - see Tn3Geom): 2 / 6 / 8 / 8 tiles cover the whole output, so ONE group of that many workgroups reads
//     every dY / X row of its contraction slice exactly once between them,
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
// Measured on MI355X (131072 rows, tools/tn384_lab.py, tools/tn384_pmc.sh; docs/LAB_NOTEBOOK.md section 4b): FETCH_SIZE = 503.6 MB for fc1 =
// 1.00 x its algorithmic bytes, 0 LDS bank-conflict cycles; main loop 1.0 PFLOP/s (1.4 with the DMA removed - the practical
// bf16 ceiling of this power-limited board is ~1.25), the same for L2-, MALL- and HBM-resident operands; the atomic epilogue of
// 288 KiB per workgroup costs 40-55 us per launch, which is why the engine launches the products in PAIRS (ccd_gemm_tn_pair).
#pragma once

namespace ccd {

constexpr int TN3_BK = 32;                                   // contraction rows per stage
// Geometry of a workgroup: WM x WN waves of 96 x 96 outputs, STAGES LDS buffers (STAGES - 1 stages of DMA in flight).
//   <4, 2, 4>: 384 x 192 tile, 8 waves, 4 x 36 KiB, one workgroup per CU
//   <4, 2, 3, 4, 2>: 512 x 128 tile (128 x 64 per wave), 8 waves, 3 x 40 KiB: the E = 512 shapes (vit_base: 1536 / 512 / 2048 x 512,
//              512 x 2048), which 384 x 192 tiles do not divide.  MEASURED on vit_base (B = 128, 65536 rows): 44.6 against 44.3 ms per step
//              with the 128-square kernel - opt-in (policy gemm_tn384_geom = 2), tested
//   (a 192 x 192 tile with two 4-wave workgroups per CU was measured in round 2 - MLP pair 0.333 vs 0.327 ms, 56.7 vs 54.3 ms per
//   step - and removed from the product in round 3)
template <int WM, int WN, int STAGES_, int TI_ = 3, int TJ_ = 3>
struct Tn3Geom {
    static constexpr int WAVES = WM * WN, THREADS = 64 * WAVES, STAGES = STAGES_, TI = TI_, TJ = TJ_;   // TI x TJ MFMA tiles per wave
    static constexpr int TP = 32 * TI * WM, TQ = 32 * TJ * WN;
    static constexpr int A_ROWB = TP * 2, B_ROWB = TQ * 2;
    static constexpr int A_BYTES = TN3_BK * A_ROWB, B_BYTES = TN3_BK * B_ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int SMEM_BYTES = STAGES * STAGE_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024 / WAVES;                  // 1-KiB DMA pieces per wave and stage (3 or 4)
    static constexpr int B_TOTAL = B_BYTES / 1024;                          // B pieces per stage: 12 (8)
    static constexpr int B_PIECES = (B_TOTAL + WAVES - 1) / WAVES;          // per wave: 2 (the second only in waves 0-3), 3 or 1
    static constexpr int B_FULL = B_TOTAL - (B_PIECES - 1) * WAVES;         // waves that own a last B piece
    static constexpr int PER_STAGE = A_PIECES + B_PIECES;
    static_assert((TI == 3 && TJ == 3) || (TI == 4 && TJ == 2), "the two hand-scheduled stage bodies below");
    static_assert(A_BYTES % (1024 * WAVES) == 0 && (PER_STAGE == 5 || PER_STAGE == 6), "piece schedule below");
};
// 16-byte chunk swizzle of an image whose rows are ROWB bytes: the four rows of one transposing read must fall on four
// different 64-byte bank groups (rows 0 mod 256 apart: all four collide; 128 mod 256: rows r and r + 2 collide)
template <int ROWB>
__device__ __forceinline__ int tn3_swz(int row) {
    static_assert(ROWB % 256 == 0 || ROWB % 256 == 128, "");
    return ROWB % 256 == 0 ? (row & 3) << 2 : ((row >> 1) & 1) << 2;
}

// p.M = P, p.N = Q, p.K = contraction length, p.k_per_split = rows per slice (multiple of 32), p.work_items = number of slices,
// p.m_fastest = XCDs the grid is spread over (8, or 1: every workgroup is its own group member in launch order).
// p.M2 > 0: a second problem (A2, B2, C2; same K; slices2 slices of per2 rows) in the same launch - which then pays ONE atomic
// epilogue per workgroup for two products (the epilogue, ~55 us, is a third of a single product's time).
// Grid: xcds * slots; workgroup b -> xcd b % xcds, slot b / xcds.  The slots of XCD x: units1[x] groups of the first problem's
// tiles, then units2[x] groups of the second's; the k-th group of a problem (counted over the XCDs) works on its slice k.  The two
// problems' groups are placed independently, so an XCD with 31 usable slots (one CU reserved for the collectives) still holds
// three groups of 8 (the host alternates 2 + 1 and 1 + 2) instead of one pair of 16.
template <int WM, int WN, int STAGES, int TI = 3, int TJ = 3>
__global__ __launch_bounds__(64 * WM * WN, 8 / (WM * WN)) void gemm_tn384_kernel(GemmParams p) {
    using G = Tn3Geom<WM, WN, STAGES, TI, TJ>;
    constexpr int AHEAD = STAGES - 1;                        // stages of DMA in flight
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hf = lane >> 5, lq = lane & 31;
    const int wm = w % WM, wn = w / WM;
    const int tiles1 = (p.M / G::TP) * (p.N / G::TQ), tiles2 = (p.M2 / G::TP) * (p.N2 / G::TQ);
    const int xcds = p.m_fastest, xcd = (int)blockIdx.x % xcds, slot = (int)blockIdx.x / xcds;
    const int u1 = (int)((p.units1 >> (8 * xcd)) & 255ull), u2 = (int)((p.units2 >> (8 * xcd)) & 255ull);
    int base1 = 0, base2 = 0;                                // groups of the XCDs in front of this one
    for (int y = 0; y < xcd; ++y) {
        base1 += (int)((p.units1 >> (8 * y)) & 255ull);
        base2 += (int)((p.units2 >> (8 * y)) & 255ull);
    }
    int tile, slice;
    if (slot < u1 * tiles1) {
        tile = slot % tiles1;
        slice = base1 + slot / tiles1;
    } else {                                                 // a tile of the second problem (wave-uniform)
        const int s2 = slot - u1 * tiles1;
        if (s2 >= u2 * tiles2) return;
        tile = s2 % tiles2;
        slice = base2 + s2 / tiles2;
        p.A = p.A2; p.B = p.B2; p.lda = p.lda2; p.ldb = p.ldb2; p.M = p.M2; p.N = p.N2; p.C = p.C2; p.ldc = p.ldc2;
        p.k_per_split = p.per2; p.work_items = p.slices2; p.ws = p.ws2;
    }
    if (slice >= p.work_items) return;
    const int tiles_q = p.N / G::TQ;
    const int p0 = (tile / tiles_q) * G::TP, q0 = (tile % tiles_q) * G::TQ;
    // (Unequal slices - lengths per + d (2 s - S + 1), so that the workgroups reach their fp32-atomic epilogues at different times - were
    // measured in round 3: MLP pair 0.337 -> 0.327 ms at +- 8-14 %, worse beyond; attention pair 0.201 -> 0.182 at +- 36 %; nothing in the
    // step.  profiles/r03_tn384_skew_lab.jsonl.)
    const int k_begin = slice * p.k_per_split;
    const int k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
    const int nk = k_end > k_begin ? (k_end - k_begin) / TN3_BK : 0;
    if (nk == 0) return;

    // ---- DMA sources: piece n of a part covers LDS bytes [1024 n, 1024 n + 1024) of that part, lane L its 16-byte chunk L.
    // A: wave w moves pieces A_PIECES w ..; B: pieces w, w + WAVES, ... (the last one only in waves < B_FULL)
    const bf16_t* ga[G::A_PIECES];
    const bf16_t* gb[G::B_PIECES];
#pragma unroll
    for (int i = 0; i < G::A_PIECES; ++i) {
        const int byte = 1024 * (G::A_PIECES * w + i) + 16 * lane;
        const int row = byte / G::A_ROWB, pos = (byte % G::A_ROWB) >> 4;
        ga[i] = p.A + (long)(k_begin + row) * p.lda + p0 + (pos ^ tn3_swz<G::A_ROWB>(row)) * 8;
    }
#pragma unroll
    for (int i = 0; i < G::B_PIECES; ++i) {
        const int byte = 1024 * (w + G::WAVES * i) + 16 * lane;
        const int row = (byte / G::B_ROWB) & (TN3_BK - 1), pos = (byte % G::B_ROWB) >> 4;
        gb[i] = p.B + (long)(k_begin + row) * p.ldb + q0 + (pos ^ tn3_swz<G::B_ROWB>(row)) * 8;
    }
    const long a_step = (long)TN3_BK * p.lda, b_step = (long)TN3_BK * p.ldb;
    const bool b_last = w < G::B_FULL;                       // this wave owns a piece of the last (partial) round of B pieces
    // piece i of stage s: first the wave's A pieces, then its B pieces.  Called for s = 0, 1, 2, ... (the sources advance).  In
    // the main loop the pieces go out BETWEEN the MFMA rows of a stage: a CU accepts one 1-KiB piece per ~17 cycles (tools/probe/
    // ldsdma_bw.hip: 60 B / cycle), issued back to back behind the barrier they kept every wave ~600 cycles in VMEM issue.
    auto dma_piece = [&](int s, int i) {
        char* base = smem + (s % STAGES) * G::STAGE_BYTES;
        if (i < G::A_PIECES) {
            stream_glds16<NT_TN>(ga[i], base + (G::A_PIECES * w + i) * 1024);
            ga[i] += a_step;
        } else {
            const int ib = i - G::A_PIECES;
            if (ib < G::B_PIECES - 1 || b_last) {
                stream_glds16<NT_TN>(gb[ib], base + G::A_BYTES + (w + G::WAVES * ib) * 1024);
                gb[ib] += b_step;
            }
        }
    };
    auto dma = [&](int s) {
#pragma unroll
        for (int i = 0; i < G::PER_STAGE; ++i) dma_piece(s, i);
    };
    // Counted waits see ONLY LDS-DMA operations (PER_STAGE per stage, one less in the waves without a last B piece).  Loads that
    // return to VGPRs must not be mixed into the window: measured here - with 1-byte L2-prefetch loads (and out-of-range buffer
    // loads as padding) between the DMA pieces, `vmcnt(N)` was satisfied by the fast VGPR returns while older DMA pieces were
    // still in flight (NaNs).  (That prefetch - also as 4-byte LDS-DMAs, which keep the order - made the loop 5-10 % SLOWER, and
    // the time per stage is the same for L2-, MALL- and HBM-resident operands: the loop is not waiting for memory.)
    auto wait_steps = [&](int in_flight) {                   // at most `in_flight` stages of this wave's DMA still outstanding
        if (b_last) {
            if (in_flight >= 3) glds_wait<3 * G::PER_STAGE>();
            else if (in_flight == 2) glds_wait<2 * G::PER_STAGE>();
            else if (in_flight == 1) glds_wait<G::PER_STAGE>();
            else glds_wait_all();
        } else {
            if (in_flight >= 3) glds_wait<3 * (G::PER_STAGE - 1)>();
            else if (in_flight == 2) glds_wait<2 * (G::PER_STAGE - 1)>();
            else if (in_flight == 1) glds_wait<G::PER_STAGE - 1>();
            else glds_wait_all();
        }
    };

    // ---- fragment addresses (see the header): lane (j = lane & 15, half-row group g16, hf)
    const int j = lane & 15, g16 = (lane >> 4) & 1, r4 = j >> 2, c4 = j & 3;
    unsigned base_a[TI], base_b[TJ];
    {
        const int row = 8 * hf + r4;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int chunk_a = 4 * TI * wm + 4 * i + 2 * g16 + (c4 >> 1);
            base_a[i] = (unsigned)(row * G::A_ROWB + ((chunk_a ^ tn3_swz<G::A_ROWB>(row)) << 4) + (c4 & 1) * 8);
        }
#pragma unroll
        for (int i = 0; i < TJ; ++i) {
            const int chunk_b = 4 * TJ * wn + 4 * i + 2 * g16 + (c4 >> 1);
            base_b[i] = (unsigned)(G::A_BYTES + row * G::B_ROWB + ((chunk_b ^ tn3_swz<G::B_ROWB>(row)) << 4) + (c4 & 1) * 8);
        }
    }

    const unsigned smem_addr = lds_addr_of(smem);
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

#ifdef CCD_MLP_LAB    // per-phase cycle totals of the first and the last wave (lab bit 4): -> p.colsum_a, 6 u64 per (workgroup < 32, wave)
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define TN3_STAMP(i) if (p.rps_shift & 4) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tprev; tprev = tn_; }
#else
#define TN3_STAMP(i)
#endif
#pragma unroll
    for (int s = 0; s < AHEAD; ++s)
        if (s < nk) dma(s);
    wait_steps((nk < AHEAD ? nk : AHEAD) - 1);
    lds_barrier();
    for (int kt = 0; kt < ((p.rps_shift & 2) ? 1 : nk); ++kt) {      // (lab bit 2: one stage only)
        const bool more = kt + AHEAD < nk && !(p.rps_shift & 8);      // (lab bit 8: no DMA inside the loop)
        //                   // stage kt + AHEAD goes into the buffer read in step kt - 1 (every
        TN3_STAMP(0)                                         // wave has passed that step's barrier)
        // Fragment reads by hand (prelude: lds_read_tr), 4 per group, LDS returns in order: group order a0 b0 | b1 b2 | a1 a2 of
        // k-step 0, then the same of k-step 1 issued between the MFMA rows of k-step 0; every wait names the youngest fragment
        // its MFMAs need and the number of reads issued behind it.
        const unsigned sb = smem_addr + (unsigned)((kt % STAGES) * G::STAGE_BYTES);
        const unsigned aa0 = sb + base_a[0], aa1 = sb + base_a[1], aa2 = sb + base_a[2], aa3 = sb + base_a[TI - 1];
        const unsigned ab0 = sb + base_b[0], ab1 = sb + base_b[1], ab2 = sb + base_b[TJ - 1];
        tr_u32x2 x[2][6][2];                                  // [k-step][a0 b0 b1 b2 a1 a2][rows 0-3 / 4-7]
#define TN3_RD_A(ks, slot, addr)                                                     \
        lds_read_tr<(16 * ks) * G::A_ROWB>(x[ks][slot][0], addr);                       \
        lds_read_tr<(16 * ks + 4) * G::A_ROWB>(x[ks][slot][1], addr);
#define TN3_RD_B(ks, slot, addr)                                                     \
        lds_read_tr<(16 * ks) * G::B_ROWB>(x[ks][slot][0], addr);                       \
        lds_read_tr<(16 * ks + 4) * G::B_ROWB>(x[ks][slot][1], addr);
#define TN3_FRAG(ks, slot) frag_from_tr(x[ks][slot][0], x[ks][slot][1])
        if constexpr (TI == 3) {
        TN3_RD_A(0, 0, aa0) TN3_RD_B(0, 1, ab0) TN3_RD_B(0, 2, ab1) TN3_RD_B(0, 3, ab2) TN3_RD_A(0, 4, aa1) TN3_RD_A(0, 5, aa2)
        bf16x8 a0 = TN3_FRAG(0, 0), b0 = TN3_FRAG(0, 1), b1 = TN3_FRAG(0, 2), b2 = TN3_FRAG(0, 3), a1 = TN3_FRAG(0, 4), a2 = TN3_FRAG(0, 5);
        lds_wait_frag<8>(b0);
        TN3_STAMP(1)
        acc[0][0] = mfma_32x32x16_bf16(a0, b0, acc[0][0]);
        lds_wait_frag<6>(b1);
        acc[0][1] = mfma_32x32x16_bf16(a0, b1, acc[0][1]);
        lds_wait_frag<4>(b2);
        acc[0][2] = mfma_32x32x16_bf16(a0, b2, acc[0][2]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 0);
        CCD_SCHED_FENCE();
        TN3_RD_A(1, 0, aa0) TN3_RD_B(1, 1, ab0)
        lds_wait_frag<6>(a1);
        acc[1][0] = mfma_32x32x16_bf16(a1, b0, acc[1][0]);
        acc[1][1] = mfma_32x32x16_bf16(a1, b1, acc[1][1]);
        acc[1][2] = mfma_32x32x16_bf16(a1, b2, acc[1][2]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 1);
        CCD_SCHED_FENCE();
        TN3_RD_B(1, 2, ab1) TN3_RD_B(1, 3, ab2)
        lds_wait_frag<8>(a2);
        acc[2][0] = mfma_32x32x16_bf16(a2, b0, acc[2][0]);
        acc[2][1] = mfma_32x32x16_bf16(a2, b1, acc[2][1]);
        acc[2][2] = mfma_32x32x16_bf16(a2, b2, acc[2][2]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 2);
        CCD_SCHED_FENCE();
        TN3_RD_A(1, 4, aa1) TN3_RD_A(1, 5, aa2)
        a0 = TN3_FRAG(1, 0); b0 = TN3_FRAG(1, 1); b1 = TN3_FRAG(1, 2); b2 = TN3_FRAG(1, 3); a1 = TN3_FRAG(1, 4); a2 = TN3_FRAG(1, 5);
        lds_wait_frag<8>(b0);
        acc[0][0] = mfma_32x32x16_bf16(a0, b0, acc[0][0]);
        lds_wait_frag<6>(b1);
        acc[0][1] = mfma_32x32x16_bf16(a0, b1, acc[0][1]);
        lds_wait_frag<4>(b2);
        acc[0][2] = mfma_32x32x16_bf16(a0, b2, acc[0][2]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 3);
        CCD_SCHED_FENCE();
        lds_wait_frag<2>(a1);
        acc[1][0] = mfma_32x32x16_bf16(a1, b0, acc[1][0]);
        acc[1][1] = mfma_32x32x16_bf16(a1, b1, acc[1][1]);
        acc[1][2] = mfma_32x32x16_bf16(a1, b2, acc[1][2]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 4);
        CCD_SCHED_FENCE();
        lds_wait_frag<0>(a2);
        acc[2][0] = mfma_32x32x16_bf16(a2, b0, acc[2][0]);
        acc[2][1] = mfma_32x32x16_bf16(a2, b1, acc[2][1]);
        acc[2][2] = mfma_32x32x16_bf16(a2, b2, acc[2][2]);
        } else {
        // 4 x 2 tiles per wave: group order a0 b0 | b1 a1 | a2 a3
        TN3_RD_A(0, 0, aa0) TN3_RD_B(0, 1, ab0) TN3_RD_B(0, 2, ab1) TN3_RD_A(0, 3, aa1) TN3_RD_A(0, 4, aa2) TN3_RD_A(0, 5, aa3)
        bf16x8 a0 = TN3_FRAG(0, 0), b0 = TN3_FRAG(0, 1), b1 = TN3_FRAG(0, 2), a1 = TN3_FRAG(0, 3), a2 = TN3_FRAG(0, 4), a3 = TN3_FRAG(0, 5);
        lds_wait_frag<8>(b0);
        TN3_STAMP(1)
        acc[0][0] = mfma_32x32x16_bf16(a0, b0, acc[0][0]);
        lds_wait_frag<6>(b1);
        acc[0][1] = mfma_32x32x16_bf16(a0, b1, acc[0][1]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 0);
        CCD_SCHED_FENCE();
        TN3_RD_A(1, 0, aa0) TN3_RD_B(1, 1, ab0)
        lds_wait_frag<8>(a1);                                // (4 of k-step 0 + 4 of k-step 1 behind it)
        acc[1][0] = mfma_32x32x16_bf16(a1, b0, acc[1][0]);
        acc[1][1] = mfma_32x32x16_bf16(a1, b1, acc[1][1]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 1);
        CCD_SCHED_FENCE();
        TN3_RD_B(1, 2, ab1) TN3_RD_A(1, 3, aa1)
        lds_wait_frag<10>(a2);
        acc[2][0] = mfma_32x32x16_bf16(a2, b0, acc[2][0]);
        acc[2][1] = mfma_32x32x16_bf16(a2, b1, acc[2][1]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 2);
        CCD_SCHED_FENCE();
        TN3_RD_A(1, 4, aa2) TN3_RD_A(1, 5, aa3)
        lds_wait_frag<12>(a3);
        acc[3][0] = mfma_32x32x16_bf16(a3, b0, acc[3][0]);
        acc[3][1] = mfma_32x32x16_bf16(a3, b1, acc[3][1]);
        a0 = TN3_FRAG(1, 0); b0 = TN3_FRAG(1, 1); b1 = TN3_FRAG(1, 2); a1 = TN3_FRAG(1, 3); a2 = TN3_FRAG(1, 4); a3 = TN3_FRAG(1, 5);
        lds_wait_frag<8>(b0);
        acc[0][0] = mfma_32x32x16_bf16(a0, b0, acc[0][0]);
        lds_wait_frag<6>(b1);
        acc[0][1] = mfma_32x32x16_bf16(a0, b1, acc[0][1]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 3);
        CCD_SCHED_FENCE();
        lds_wait_frag<4>(a1);
        acc[1][0] = mfma_32x32x16_bf16(a1, b0, acc[1][0]);
        acc[1][1] = mfma_32x32x16_bf16(a1, b1, acc[1][1]);
        CCD_SCHED_FENCE();
        if (more) dma_piece(kt + AHEAD, 4);
        CCD_SCHED_FENCE();
        lds_wait_frag<2>(a2);
        acc[2][0] = mfma_32x32x16_bf16(a2, b0, acc[2][0]);
        acc[2][1] = mfma_32x32x16_bf16(a2, b1, acc[2][1]);
        lds_wait_frag<0>(a3);
        acc[3][0] = mfma_32x32x16_bf16(a3, b0, acc[3][0]);
        acc[3][1] = mfma_32x32x16_bf16(a3, b1, acc[3][1]);
        }
#undef TN3_RD_A
#undef TN3_RD_B
#undef TN3_FRAG
        if (G::PER_STAGE == 6) {
            CCD_SCHED_FENCE();
            if (more) dma_piece(kt + AHEAD, 5);
        }
        TN3_STAMP(2)
        const int rem = nk - 1 - kt;                         // stage kt + 1 must have landed before the barrier publishes it
        if (rem > 0) wait_steps((rem < AHEAD ? rem : AHEAD) - 1);
        TN3_STAMP(3)
        if (!(p.rps_shift & 16)) lds_barrier();              // (lab bit 16: no barrier)
        TN3_STAMP(4)
    }
#ifdef CCD_MLP_LAB
    if ((p.rps_shift & 4) && p.colsum_a && blockIdx.x < 32 && lane == 0 && (w == 0 || w == G::WAVES - 1)) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.colsum_a) + (blockIdx.x * 2 + (w ? 1 : 0)) * 6;
        for (int i = 0; i < 6; ++i) o[i] = ph[i];
    }
#endif
    glds_wait_all();

    // ---- epilogue: D[p][q], a lane owns column q = lq of 16 rows per tile; 32 lanes = 128 contiguous bytes per atomic
    float* C = reinterpret_cast<float*>(p.C);
    if (p.rps_shift & 1) return;                             // (lab bit 1: no epilogue)
    if (p.ws) {
        // round 4: this slice's tile leaves by PLAIN stores into its own plane of the workspace (a half wave = 128 contiguous bytes
        // per store); tn3_reduce_kernel sums the planes into C.  The atomic form below moved 75 MB of fp32 atomics per MLP pair at
        // ~1.4 TB/s (40 - 55 us per launch); stores + one reduction pass move the same bytes twice at streaming rate.
        float* W = p.ws + (long)slice * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int jj = 0; jj < TJ; ++jj) {
                float* cp = W + (long)(p0 + 32 * TI * wm + 32 * i + 4 * hf) * p.N + (q0 + 32 * TJ * wn + 32 * jj + lq);
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[(long)((r & 3) + 8 * (r >> 2)) * p.N] = acc[i][jj][r];
            }
        return;
    }
    // (Rotating the order of the 3 x 3 sub-tiles by the slice number, so that concurrent slices add onto different cache lines,
    // changed nothing - 0.1914 vs 0.1905 ms: the epilogue is bound by the L2's atomic throughput, not by same-line contention.)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
            float* cp = C + (long)(p0 + 32 * TI * wm + 32 * i + 4 * hf) * p.ldc + (q0 + 32 * TJ * wn + 32 * jj + lq);
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(cp + (long)((r & 3) + 8 * (r >> 2)) * p.ldc, acc[i][jj][r] * p.alpha);
        }
}

// C[P, Q] += alpha * sum over the S slice planes of ws ([S][P][Q] fp32, what gemm_tn384_kernel's workspace epilogue wrote), for up
// to two problems in one launch: thread -> one float4 of the first problem's output, then of the second's.  16 planes x 16 B in
// flight per thread; the planes were written a few microseconds ago (L2 / MALL resident where they fit).
struct Tn3ReduceParams {
    const float* ws[2];
    float* C[2];
    long ldc[2];
    int S[2], P[2], Q[2];
    float alpha;
};
__global__ __launch_bounds__(256) void tn3_reduce_kernel(Tn3ReduceParams p) {
    const long n0 = (long)p.P[0] * p.Q[0] / 4, n1 = (long)p.P[1] * p.Q[1] / 4;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int which = 0;
    if (i >= n0) { i -= n0; which = 1; if (i >= n1) return; }
    const long plane = (long)p.P[which] * p.Q[which];
    const float* src = p.ws[which] + 4 * i;
    f32x4v s = {0.f, 0.f, 0.f, 0.f};
    const int S = p.S[which];
    int k = 0;
    for (; k + 4 <= S; k += 4) {
        const f32x4v a = *reinterpret_cast<const f32x4v*>(src + (long)k * plane), b = *reinterpret_cast<const f32x4v*>(src + (long)(k + 1) * plane);
        const f32x4v c = *reinterpret_cast<const f32x4v*>(src + (long)(k + 2) * plane), d = *reinterpret_cast<const f32x4v*>(src + (long)(k + 3) * plane);
        s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y); s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; k < S; ++k) {
        const f32x4v a = *reinterpret_cast<const f32x4v*>(src + (long)k * plane);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    const long e = 4 * i, row = e / p.Q[which], col = e % p.Q[which];
    float* dst = p.C[which] + row * p.ldc[which] + col;
    f32x4v o = *reinterpret_cast<f32x4v*>(dst);
    o.x += p.alpha * s.x; o.y += p.alpha * s.y; o.z += p.alpha * s.z; o.w += p.alpha * s.w;
    *reinterpret_cast<f32x4v*>(dst) = o;
}

}  // namespace ccd
