// gemm256.h - the large-M NT GEMM: 256x256 output tile, 8 waves (2 x 4, 128x64 per wave), BK = 64, operands staged
// by LDS-DMA (global_load_lds, prelude: glds16) into two 64-KiB LDS buffers, one workgroup per CU.
//
// Why a second NT kernel next to gemm.h's 128x128 tile: measured on MI355X (profiles/) the 128^2
// structure tops out near 850 TFLOP/s - per MFMA it moves too many bytes through the CU's single memory pipe and the
// LDS (both ~30 % busy, serialised with the MFMA phase), and its register-staged operands cost a ds_write pass and
// 32 VGPRs.  Here a wave owns 4 x 2 accumulator tiles (6 fragment reads per 8 MFMAs instead of 4 per 4), a k-tile of
// 64 KiB feeds 4x the MFMAs of the 128^2 tile's 32 KiB, and the DMA writes LDS without touching VGPRs.
//
// LDS image: rows of 128 B (64 bf16), 1-KiB chunks of 8 rows written lane-linearly by one wave instruction; the
// 16-byte slot swizzle of gemm.h (gemm_swz) is applied on the SOURCE address (lane (r, p) fetches slot p ^ f(r)), so the
// fragment reads stay conflict-free.  Rows beyond the matrix are clamped to the last valid row (their products land in
// rows / columns the epilogue never stores).
//
// Ordering: the DMA of k-tile t+1 is issued at the top of iteration t into the buffer last read in iteration t-1 (all
// its ds_reads retired before that iteration's closing barrier); `s_waitcnt vmcnt(0)` + barrier at the bottom of
// iteration t publish it.  Epilogue: four 64-row passes through a 64-KiB staging image in the second buffer (bf16 for
// the bf16 / GELU-pair outputs, fp32 otherwise), then the same fused row epilogues as gemm.h (bias / GELU pair /
// residual + DropPath / fp32 / gelu'(u) with column sums); the NEXT tile's first k-tile already streams into buffer 0.
#pragma once

namespace ccd {

constexpr int G256_BM = 256, G256_BN = 256, G256_BK = 64, G256_THREADS = 512;
constexpr int G256_OPERAND_BYTES = G256_BM * G256_BK * 2;            // 32 KiB per operand and buffer
constexpr int G256_SMEM_BYTES = 4 * G256_OPERAND_BYTES;             // 2 buffers x (A + B) = 128 KiB
constexpr int G256_MAX_COLSUM_N = 6144;                              // (+ 24 KiB of column sums at most: 152 of 160 KiB)
__host__ __device__ inline int g256_smem_bytes(int N, bool colsum) { return G256_SMEM_BYTES + (colsum ? N * 4 : 0); }

// BN = 256: waves 2 (M) x 4 (N), 128x64 per wave.  BN = 128 (the N = 384 products): waves 4 x 2, 64x64 per wave.
// DEEP = true: BK = 32 and FOUR 32-KiB buffers instead of BK = 64 and two 64-KiB ones - three k-steps of DMA in flight
// (counted vmcnt, LDS-only barriers) instead of one.  Measured on the plain variant: of ~4.0 k cycles per 64-deep k-tile
// only ~2.2 k are MFMA issue, ~1.8 k are spent waiting for the single in-flight DMA (own vmcnt + the other waves').
template <int EPI, int BN = 256, bool DEEP = false>
__global__ __launch_bounds__(G256_THREADS, 1) void gemm256_kernel(GemmParams p) {
    constexpr int WM = BN == 256 ? 2 : 4, WN = 8 / WM;     // wave grid
    constexpr int WROWS = G256_BM / WM, WCOLS = BN / WN;    // per-wave output
    constexpr int TI = WROWS / 32, TJ = WCOLS / 32;         // 32x32 accumulator tiles per wave
    constexpr int BCH = BN / 64;                            // B chunks (8 rows) per wave and k-tile
    constexpr int BK = DEEP ? 32 : 64;                      // contraction depth of one LDS buffer
    constexpr int ROW = BK * 2;                             // bytes per image row
    constexpr int A_IMG = G256_BM * ROW;                    // A image bytes inside a buffer (B follows)
    constexpr int BUF_SHIFT = DEEP ? 15 : 16;               // buffer b starts at b << BUF_SHIFT
    const int m_static = p.M;                               // the work list is built for the static shape
    if (p.d_rows) {                                         // device-side row count: tiles past it are skipped
        const int dyn = p.d_rows[0] * p.rows_mul;
        p.M = dyn < p.M ? dyn : p.M;
    }
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hf = lane >> 5, lq = lane & 31;
    // wave w runs on SIMD w % 4: with wm fastest, the waves of any column sub-range cover all four SIMDs, so a column-
    // partial tile (N = 384: 256 + 128) whose right-hand waves have nothing to multiply finishes in half the time
    const int wm = w % WM, wn = w / WM;
    const int tiles_m = (p.M + G256_BM - 1) / G256_BM, tiles_n = (p.N + BN - 1) / BN;   // live row tiles only
    const unsigned total = (unsigned)(tiles_m * tiles_n), G = gridDim.x;
    // XCD-partitioned work list (see gemm.h): consecutive tiles = the column tiles of one A row-panel
    const unsigned ng = G < 8u ? G : 8u;
    const unsigned xcd = blockIdx.x % ng, slot = blockIdx.x / ng;
    const unsigned nx = G / ng + (xcd < G % ng ? 1u : 0u);
    const unsigned q8 = total / ng, r8 = total % ng;
    const unsigned base_x = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned cnt_x = q8 + (xcd < r8 ? 1u : 0u);
    const int nk = p.K / BK;

    // Fragment addresses.  With slot = 2*kk + hf and the swizzle slot ^ f(row), the byte offset of k-step kk is
    // base ^ (kk << 5) with base = row*128 + ((f >> 1) << 5) + ((hf ^ (f & 1)) << 4): ONE register per fragment row
    // instead of one per (row, kk); the buffer bit (1 << 16) is folded into the same XOR.
    unsigned base_a[TI], base_b[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int row = WROWS * wm + 32 * i + lq;
        const int f = DEEP ? ((row >> 2) & 3) : (((row >> 1) ^ (row >> 4)) & 7);     // 64-B rows: 4 slots, 128-B rows: 8
        base_a[i] = (unsigned)(row * ROW + ((f >> 1) << 5) + ((hf ^ (f & 1)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int row = WCOLS * wn + 32 * j + lq;
        const int f = DEEP ? ((row >> 2) & 3) : (((row >> 1) ^ (row >> 4)) & 7);
        base_b[j] = (unsigned)(A_IMG + row * ROW + ((f >> 1) << 5) + ((hf ^ (f & 1)) << 4));
    }

    if (slot >= cnt_x) return;
    // (Round 4 measured a phase skew here - workgroup slot s starting (s % phases) * delay late, 2 / 4 / 8 phases, 8 - 32 k cycles -
    // on the theory that 256 CUs storing their epilogues at once saturate HBM while the main loops leave it idle: 0.349 - 0.352
    // against 0.355 ms for the gelu' product, nothing.  The workgroups are not phase-locked; see docs/LAB_NOTEBOOK.md section 4e.)
    // column sums (bias gradients): LDS accumulators behind the operand buffers for every column of the product, flushed by
    // one pass of global atomics when the workgroup is done (was: per tile a transpose through the staging image, three
    // barriers and 256 global atomics - 40 us of a 360-us launch).  The launcher sizes the LDS (g256_smem_bytes).
    float* cs_lds = reinterpret_cast<float*>(smem + G256_SMEM_BYTES);
    const bool cs_on = (EPI == EPI_DGELU || EPI == EPI_BF16) && p.colsum != nullptr;
    if (cs_on) {
        for (int i = t; i < p.N; i += G256_THREADS) cs_lds[i] = 0.f;
        __syncthreads();                                        // (nothing in flight yet)
    }
    // DMA source pointers of the current item: wave w moves chunks 4w .. 4w+3 (8 rows each) of both operands
    // a 1-KiB DMA chunk = 8 rows of 128 B (BK = 64) or 16 rows of 64 B (BK = 32); wave w moves ACH / BCHK chunks per buffer
    constexpr int RPC = 1024 / ROW;                         // rows per chunk
    constexpr int ACH = G256_BM / RPC / 8, BCHK = BN / RPC / 8;
    const bf16_t* ga[ACH];
    const bf16_t* gb[BCHK];
    int m0, n0;
    bool live;
    auto setup = [&](unsigned item) {
        const unsigned tile = base_x + item;
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        m0 = tm * G256_BM;
        n0 = tn * BN;
        live = m0 < p.M;
        const int rin = DEEP ? (lane >> 2) : (lane >> 3), pos = DEEP ? (lane & 3) : (lane & 7);
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int row = RPC * (ACH * w + i) + rin;
            const int src_slot = pos ^ (DEEP ? ((row >> 2) & 3) : (((row >> 1) ^ (row >> 4)) & 7));
            int ra = m0 + row;
            ra = ra < p.M ? ra : p.M - 1;
            ga[i] = p.A + (long)ra * p.lda + src_slot * 8;
        }
#pragma unroll
        for (int i = 0; i < BCHK; ++i) {
            const int row = RPC * (BCHK * w + i) + rin;
            const int src_slot = pos ^ (DEEP ? ((row >> 2) & 3) : (((row >> 1) ^ (row >> 4)) & 7));
            int rb = n0 + row;
            rb = rb < p.N ? rb : p.N - 1;
            gb[i] = p.B + (long)rb * p.ldb + src_slot * 8;
        }
    };
    auto dma = [&](int kt, int buf) {
        char* abuf = smem + (buf << BUF_SHIFT) + ACH * w * 1024;
        char* bbuf = smem + (buf << BUF_SHIFT) + A_IMG + BCHK * w * 1024;
#pragma unroll
        for (int i = 0; i < ACH; ++i) glds16(ga[i] + kt * BK, abuf + i * 1024);
#pragma unroll
        for (int i = 0; i < BCHK; ++i) glds16(gb[i] + kt * BK, bbuf + i * 1024);
    };
#ifdef CCD_GEMM_LAB     // per-phase cycle totals of wave 0 -> p.colsum (8 u64 per workgroup) when m_fastest & 64
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define G256_STAMP(i) if (p.m_fastest & 64) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tprev; tprev = tn_; }
#else
#define G256_STAMP(i)
#endif
    // DEEP: wait until at most `tiles_in_flight` k-steps of DMA (ACH + BCHK pieces each) are still outstanding
    auto wait_steps = [&](int tiles_in_flight) {
        constexpr int PER = ACH + BCHK;
        if (tiles_in_flight >= 2) glds_wait<2 * PER>();
        else if (tiles_in_flight == 1) glds_wait<PER>();
        else glds_wait_all();
    };
    // bias of the tile's columns: one value per thread, requested a whole tile ahead (right after `setup`) and parked in
    // LDS at the start of the epilogue - read from global memory inside the staging loop, each 16-byte piece was a
    // serialised L2 round trip behind `s_waitcnt vmcnt(0)` (which also drained the next tile's DMA)
    constexpr bool BIAS_LDS = (EPI == EPI_BF16 || EPI == EPI_GELU);
    float bias_r = 0.f;
    auto load_bias = [&]() {
        bias_r = 0.f;
        if (BIAS_LDS && p.bias && t < BN && n0 + t < p.N) bias_r = p.bias[n0 + t];
    };
    // gelu'(u) epilogue: descriptors of the pre-activations and the two outputs (rows < M only; the ABI keeps them below 2 GiB)
    buf_rsrc r_aux = make_rsrc(nullptr, 0), r_c = make_rsrc(nullptr, 0), r_c2 = make_rsrc(nullptr, 0);
    if (EPI == EPI_DGELU) {
        r_aux = make_rsrc(p.aux, (unsigned)((((long)p.M - 1) * p.ldaux + p.N) * 2));
        r_c = make_rsrc(p.C, (unsigned)((((long)p.M - 1) * p.ldc + p.N) * 2));
        if (p.C2) r_c2 = make_rsrc(p.C2, (unsigned)((((long)p.M - 1) * p.ldc2 + p.N) * 2));
    }
    unsigned item = slot;
    setup(item);
    load_bias();
    if (live) {
        dma(0, 0);
        if (DEEP) {
            if (nk > 1) dma(1, 1);
        }
    }
    while (true) {
        f32x16 acc[TI][TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int nk_live = live ? nk : 0;
        const bool wave_live = n0 + WCOLS * wn < p.N;         // this wave's columns exist (column-partial tiles)
        if (DEEP) {
            // steps 0 and 1 were requested before (prologue / under the previous epilogue); step 2 goes out now that the
            // staging image (buffers 2, 3) has been read out.  vmcnt(0) once per tile also retires the epilogue's stores.
            if (nk_live > 2) dma(2, 2);
            wait_steps(nk_live > 2 ? 2 : (nk_live > 1 ? 1 : 0));     // step 0 landed
            lds_barrier();
            G256_STAMP(0)
            for (int kt = 0; kt < nk_live; ++kt) {
                if (kt + 3 < nk_live) dma(kt + 3, (kt + 3) & 3);      // buffer last read in step kt-1 (barrier passed)
                const unsigned bufbit = (unsigned)(kt & 3) << BUF_SHIFT;
                if (wave_live) {
                    bf16x8 a[2][TI], b[2][TJ];
#pragma unroll
                    for (int j = 0; j < TJ; ++j) b[0][j] = *reinterpret_cast<const bf16x8*>(smem + (base_b[j] ^ bufbit));
#pragma unroll
                    for (int i = 0; i < TI; ++i) a[0][i] = *reinterpret_cast<const bf16x8*>(smem + (base_a[i] ^ bufbit));
                    {
                        const unsigned x = bufbit | 32u;
#pragma unroll
                        for (int j = 0; j < TJ; ++j) b[1][j] = *reinterpret_cast<const bf16x8*>(smem + (base_b[j] ^ x));
#pragma unroll
                        for (int i = 0; i < TI; ++i) a[1][i] = *reinterpret_cast<const bf16x8*>(smem + (base_a[i] ^ x));
                    }
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < TI; ++i)
#pragma unroll
                            for (int j = 0; j < TJ; ++j)
                                acc[i][j] = mfma_32x32x16_bf16(b[kk][j], a[kk][i], acc[i][j]);   // D^T[n][m]
                }
                G256_STAMP(1)
                // step kt+1 must have landed: after this step's issue min(3, remaining) steps are in flight
                const int rem = nk_live - 1 - kt;
                if (rem > 0) wait_steps((rem < 3 ? rem : 3) - 1);
                G256_STAMP(2)
                lds_barrier();
                G256_STAMP(3)
            }
        } else {
        glds_wait_all();
        __syncthreads();
        G256_STAMP(0)
        for (int kt = 0; kt < nk_live; ++kt) {
            if (kt + 1 < nk) dma(kt + 1, (kt + 1) & 1);
            const unsigned bufbit = (unsigned)(kt & 1) << 16;       // buffer 1 starts at 64 KiB
            if (wave_live) {
#ifdef CCD_G256_PRIO
            wave_prio<1>();
#endif
            // fragments of k-step kk+1 are requested before the 8 MFMAs of k-step kk are issued
            bf16x8 a[2][TI], b[2][TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[0][j] = *reinterpret_cast<const bf16x8*>(smem + (base_b[j] ^ bufbit));
#pragma unroll
            for (int i = 0; i < TI; ++i) a[0][i] = *reinterpret_cast<const bf16x8*>(smem + (base_a[i] ^ bufbit));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk < 3) {
                    const unsigned x = bufbit | (unsigned)((kk + 1) << 5);
#pragma unroll
                    for (int j = 0; j < TJ; ++j) b[nxt][j] = *reinterpret_cast<const bf16x8*>(smem + (base_b[j] ^ x));
#pragma unroll
                    for (int i = 0; i < TI; ++i) a[nxt][i] = *reinterpret_cast<const bf16x8*>(smem + (base_a[i] ^ x));
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = mfma_32x32x16_bf16(b[cur][j], a[cur][i], acc[i][j]);   // D^T[n][m]
            }
            }
#ifdef CCD_G256_PRIO
            wave_prio<0>();
#endif
            G256_STAMP(1)
            glds_wait_all();
            G256_STAMP(2)
            __syncthreads();
            G256_STAMP(3)
        }
        }
        // ---- next item: its first k-tile streams into buffer 0 while this tile is written out through buffer 1
        const int em0 = m0, en0 = n0;
        const bool elive = live, ewave_live = wave_live;
        const unsigned next = item + nx;
        const bool has_next = next < cnt_x;
        float* bias_s = reinterpret_cast<float*>(smem + 2 * G256_OPERAND_BYTES + 45056);     // behind staging image + two tables
        if (BIAS_LDS && elive) {
            if (t < BN) bias_s[t] = bias_r;                  // (no DMA in flight here: the wait for bias_r drains nothing)
            lds_barrier();
        }
        if (has_next) {
            setup(next);
            load_bias();
            if (live) {
                dma(0, 0);
                if (DEEP) {
                    if (nk > 1) dma(1, 1);                   // buffers 0 and 1; the staging image owns buffers 2 and 3
                }
            }
        }
        if (elive) {

        // ---- epilogue (LDS-only barriers: the DMA stays in flight).  The products were accumulated TRANSPOSED, so a
        // lane owns 4 consecutive columns of one row: 16-byte (fp32) / 8-byte (bf16) staging writes.  TI passes, pass q
        // = the q-th 32-row slab of every wave = 32*WM tile rows x BN columns, in a 16-byte-chunk XOR-swizzled image
        // (at most 64 KiB: 64 x 256 or 128 x 128 fp32).
        char* stg = smem + 2 * G256_OPERAND_BYTES;
        constexpr bool STAGE_BF16 = (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_DGELU);   // bf16 outputs: staged packed
        // (DGELU: the bf16-rounded product is multiplied by gelu'(u) in the row pass, where the aux read is coalesced)
        // GELU / gelu' by table: their argument is a bf16 value (the stored pre-activation u), and Phi / gelu' need only
        // |u| in [2^-8, 16) = 1536 bf16 bit patterns (below: Phi ~ 0.5 to 6e-6 in the product, above: exactly 0 / 1;
        // f(-u) = 1 - f(u) for both).  One LDS gather + ~8 VALU per element instead of ~20-25 VALU of erf / exp:
        // the transcendental epilogue used to cost as many cycles as the K = 384 main loop.
        constexpr unsigned LUT_LO = 0x3B80u, LUT_HI = 0x4180u;
        constexpr bool USE_LUT = (EPI == EPI_GELU || EPI == EPI_DGELU);
        constexpr int SROWS = 32 * WM;                        // staged rows per pass
        constexpr int ROWB = STAGE_BF16 ? BN * 2 : BN * 4;    // bytes per staged row
        constexpr int CT = BN / 8;                            // column threads (8 columns each) in the row pass
        constexpr int RSTEP = G256_THREADS / CT;              // rows per row-pass step
        // Table layout: entry i (|u| = bf16 bit pattern LUT_LO + i) of the non-negative half, then the same 1536 entries for negative u
        // (f(-u) = 1 - f(u), folded in when the table is built): the row pass needs no compare / subtract / select per element.  The gelu'
        // epilogue keeps {gelu'(u), Phi(u)} side by side - ONE 8-byte gather per element serves both outputs.  The gathers of a row
        // step are issued back to back by hand and waited for ONCE: compiler-issued, each ds_read_b32 was followed by its own
        // `s_waitcnt lgkmcnt(0)` - 16 serialised LDS round trips per 8 elements, two thirds of this kernel's epilogue (round 4).
        constexpr unsigned LUT_N = LUT_HI - LUT_LO;
        constexpr int LUT_STRIDE = EPI == EPI_DGELU ? 8 : 4;              // bytes per entry
        constexpr int LUT_NEG = 16384;                                    // gelu' table: byte offset of the negative half (32 KiB in all)
        float* lut = reinterpret_cast<float*>(stg + SROWS * ROWB);        // behind the (single) bf16 staging image
        const bool want_gelu = EPI == EPI_DGELU && p.C2 != nullptr;
        if (USE_LUT) {
            for (unsigned i = t; i < LUT_N; i += G256_THREADS) {
                const float x = bf2f((bf16_t)(LUT_LO + i));
                const GeluTerms gt = gelu_terms(x);
                if (EPI == EPI_GELU) {
                    lut[i] = gt.cdf;
                    lut[LUT_N + i] = 1.0f - gt.cdf;
                } else {
                    const float d = fmaf(x * 0.3989422804014327f, gt.gauss, gt.cdf);
                    lut[2 * i] = d;
                    lut[2 * i + 1] = gt.cdf;
                    lut[LUT_NEG / 4 + 2 * i] = 1.0f - d;
                    lut[LUT_NEG / 4 + 2 * i + 1] = 1.0f - gt.cdf;
                }
            }
        }
        const unsigned lut_addr = lds_addr_of(lut);
        // the same for the TWO bf16 patterns of a dword, as two 16-bit byte offsets into the gelu' table (packed 16-bit VALU: and, max,
        // min, sub, shift for two elements; the negative half of that table starts LUT_NEG = 16 KiB in, so the sign bit, shifted
        // right by one, IS its offset)
        auto lut_offsets2 = [&](unsigned w) -> unsigned {
            const u16x2 lo2 = {(unsigned short)LUT_LO, (unsigned short)LUT_LO}, hi2 = {(unsigned short)(LUT_HI - 1u), (unsigned short)(LUT_HI - 1u)};
            u16x2 m = __builtin_bit_cast(u16x2, w & 0x7fff7fffu);
            m = __builtin_elementwise_min(__builtin_elementwise_max(m, lo2), hi2);
            const u16x2 off = (m - lo2) << 3;                                  // < 12 288
            return __builtin_bit_cast(unsigned, off) | ((w >> 1) & 0x40004000u);    // sign bit -> + LUT_NEG bytes
        };
        auto lut_entry = [&](unsigned bits) -> unsigned {          // LDS address of the entry of the bf16 pattern in bits[15:0]
            const unsigned mag = bits & 0x7fffu;
            const unsigned c = mag < LUT_LO ? LUT_LO : (mag > LUT_HI - 1u ? LUT_HI - 1u : mag);
            return lut_addr + (c - LUT_LO) * LUT_STRIDE + ((bits >> 15) & 1u) * (LUT_N * LUT_STRIDE);
        };
        if (p.alpha != 1.0f) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.alpha;
        }
#ifdef CCD_GEMM_LAB
        const bool want_stats = (EPI == EPI_DGELU || EPI == EPI_BF16) && p.colsum != nullptr && !(p.m_fastest & 64);
#else
        const bool want_stats = (EPI == EPI_DGELU || EPI == EPI_BF16) && p.colsum != nullptr;
#endif
        float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int ct = t % CT, rr = t / CT;                   // row pass: 8 columns per thread, RSTEP rows per step
        const int gn = en0 + 8 * ct;
        // gelu'(u) epilogue: the pre-activations of a pass's rows are requested ONE PASS AHEAD - after the previous pass's rows have
        // been waited for and before its stores go out - so their HBM latency is spent under that pass's row sweep (requested at the
        // top of their own pass they cost a round trip per pass: four per tile, a quarter of the kernel)
        u32x4 auxq[2][SROWS / RSTEP];
        // gelu'(u) epilogue: buffer addressing - the lane part (the thread's row inside a step, its 8 columns; a column past N is
        // out of range) is computed once per tile, the step's first row is a wave-uniform SGPR offset, rows past M fall outside the
        // descriptor: no 64-bit address arithmetic and no exec-mask branch per step (10 of ~150 VALU instructions per step before)
        auto step_row = [&](int q, int pass) -> int {            // first tile row of row step `pass` of pass q
            return em0 + WROWS * ((pass * RSTEP) >> 5) + 32 * q + ((pass * RSTEP) & 31);
        };
        const bool col_ok = gn < p.N;
        const unsigned v_aux = col_ok ? (unsigned)((rr * p.ldaux + gn) * 2) : BUF_OOB;
        const unsigned v_c = col_ok ? (unsigned)((rr * p.ldc + gn) * 2) : BUF_OOB;
        const unsigned v_c2 = col_ok ? (unsigned)((rr * p.ldc2 + gn) * 2) : BUF_OOB;
        auto load_aux = [&](int q, u32x4 (&dst)[SROWS / RSTEP]) {
#pragma unroll
            for (int pass = 0; pass < SROWS / RSTEP; ++pass)
                dst[pass] = stream_load16<NT_DGELU>(r_aux, v_aux, (unsigned)(step_row(q, pass) * (int)p.ldaux * 2));
        };
        if (EPI == EPI_DGELU) load_aux(0, auxq[0]);
#pragma unroll
        for (int q = 0; q < TI; ++q) {
            const int srow = 32 * wm + lq;
            u32x4 (&auxw)[SROWS / RSTEP] = auxq[q & 1];
            if (ewave_live)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = WCOLS * wn + 32 * j + 8 * g + 4 * hf;   // first of this lane's 4 columns
                    float v0 = acc[q][j][4 * g], v1 = acc[q][j][4 * g + 1], v2 = acc[q][j][4 * g + 2], v3 = acc[q][j][4 * g + 3];
                    if (STAGE_BF16) {
                        if (BIAS_LDS) {
                            const f32x4v b = *reinterpret_cast<const f32x4v*>(bias_s + nl);
                            v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
                        }
                        char* dst = stg + srow * ROWB + (((nl >> 3) ^ (srow & 15)) * 16) + ((nl >> 2) & 1) * 8;
                        u32x2 o;
                        o.x = pack_bf2(v0, v1);
                        o.y = pack_bf2(v2, v3);
                        *reinterpret_cast<u32x2*>(dst) = o;
                    } else {
                        const f32x4v o = {v0, v1, v2, v3};
                        *reinterpret_cast<f32x4v*>(stg + srow * ROWB + (((nl >> 2) ^ (srow & 15)) * 16)) = o;
                    }
                }
            G256_STAMP(4)
            lds_barrier();
            G256_STAMP(5)
            if (EPI == EPI_DGELU) {           // every row of this pass has arrived BEFORE the first store (prelude: needed_here)
#pragma unroll
                for (int pass = 0; pass < SROWS / RSTEP; ++pass) needed_here(auxw[pass]);
                if (q + 1 < TI) load_aux(q + 1, auxq[(q + 1) & 1]);
                G256_STAMP(7)
            }
            if (EPI == EPI_DGELU) {
#pragma unroll
                for (int pass = 0; pass < SROWS / RSTEP; ++pass) {
                    const int s2 = pass * RSTEP + rr;
                    const u32x4 uw = auxw[pass];
                    f32x2 tab[8];                             // {gelu'(u), Phi(u)} of the 8 pre-activations
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned o2 = lut_offsets2(uw[i]);
                        lds_gather_f32x2(tab[2 * i], lut_addr + (o2 & 0xffffu));
                        lds_gather_f32x2(tab[2 * i + 1], lut_addr + (o2 >> 16));
                    }
                    const u32x4 pw = *reinterpret_cast<const u32x4*>(stg + s2 * ROWB + ((ct ^ (s2 & 15)) * 16));
                    lds_drain();
                    lds_landed8(tab);
                    f32x2 r[8];                               // {product * gelu'(u), u * Phi(u)}: one packed multiply per element
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned pe = pw[e >> 1], ue = uw[e >> 1];
                        f32x2 x;
                        x.x = __builtin_bit_cast(float, (e & 1) ? (pe & 0xffff0000u) : (pe << 16));
                        x.y = __builtin_bit_cast(float, (e & 1) ? (ue & 0xffff0000u) : (ue << 16));
                        r[e] = x * tab[e];
                    }
                    u32x4 oc, og;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        oc[i] = pack_bf2(r[2 * i].x, r[2 * i + 1].x);
                        og[i] = pack_bf2(r[2 * i].y, r[2 * i + 1].y);
                    }
                    stream_store16<NT_DGELU>(r_c, v_c, (unsigned)(step_row(q, pass) * (int)p.ldc * 2), oc);
                    if (want_gelu) stream_store16<NT_DGELU>(r_c2, v_c2, (unsigned)(step_row(q, pass) * (int)p.ldc2 * 2), og);
                    if (want_stats) {                         // rows past M: products of clamped operand rows - not counted
                        if (step_row(q, pass) + rr < p.M) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) csum[e] += r[e].x;
                        }
                    }
                }
            } else
#pragma unroll
            for (int pass = 0; pass < SROWS / RSTEP; ++pass) {
                const int s2 = pass * RSTEP + rr;
                const int gm = em0 + WROWS * (s2 >> 5) + 32 * q + (s2 & 31);
                if (gm < p.M && gn < p.N) {
                    if (STAGE_BF16) {
                        const char* src = stg + s2 * ROWB + ((ct ^ (s2 & 15)) * 16);
                        if (EPI == EPI_BF16 || p.C) {
                            const u32x4 wv = *reinterpret_cast<const u32x4*>(src);
                            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = wv;
                            if (want_stats) {
                                float v[8];
                                unpack8(wv, v);
#pragma unroll
                                for (int e = 0; e < 8; ++e) csum[e] += v[e];
                            }
                        }
                        if (EPI == EPI_GELU) {                // gelu(u) of the bf16 pre-activation that backward will see
                            const u32x4 uw = *reinterpret_cast<const u32x4*>(src);
                            float cdf[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) lds_gather_f32(cdf[e], lut_entry(uw[e >> 1] >> (16 * (e & 1))));
                            float gq[8];
                            unpack8(uw, gq);
                            lds_drain();
                            lds_landed(cdf[0], cdf[1]); lds_landed(cdf[2], cdf[3]); lds_landed(cdf[4], cdf[5]); lds_landed(cdf[6], cdf[7]);
#pragma unroll
                            for (int e = 0; e < 8; ++e) gq[e] *= cdf[e];
                            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C2) + (long)gm * p.ldc2 + gn) = pack8(gq);
                        }
                    } else {
                        float v[8];
                        const f32x4v c0 = *reinterpret_cast<const f32x4v*>(stg + s2 * ROWB + (((2 * ct) ^ (s2 & 15)) * 16));
                        const f32x4v c1 = *reinterpret_cast<const f32x4v*>(stg + s2 * ROWB + (((2 * ct + 1) ^ (s2 & 15)) * 16));
                        v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
                        v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
                        gemm_epilogue_row8<EPI>(p, gm, gn, v);
                        if (want_stats) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) csum[e] += v[e];
                        }
                    }
                }
            }
            G256_STAMP(6)
            lds_barrier();                                   // staging image read out
            G256_STAMP(5)
        }
        if (want_stats) {
            // a wave holds every column thread twice (CT = 32: lanes l and l ^ 32 are the same 8 columns, two row threads; CT = 16:
            // four row threads, lanes l, l ^ 16, l ^ 32, l ^ 48): fold inside the wave, one LDS atomic per column and wave
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = csum[e] + shfl_xor(csum[e], 32);
                if (CT == 16) v += shfl_xor(v, 16);
                if (lane < CT && gn + e < p.N) atomicAdd(cs_lds + gn + e, v);
            }
        }
        }
        if (!has_next) break;
        item = next;
    }
    if (cs_on) {
        lds_barrier();
        // every workgroup starts its sweep at another column: 256 workgroups adding to the SAME address at the same moment
        // serialise in the L2's atomic unit (28 us of tail measured with the sweeps aligned)
        const int rot = (int)(((unsigned)blockIdx.x * 2654435761u >> 8) % (unsigned)p.N);
        for (int i = t; i < p.N; i += G256_THREADS) {
            int c = i + rot;
            c = c >= p.N ? c - p.N : c;
            const float v = cs_lds[c];
            if (v != 0.f) atomicAdd(p.colsum + c, v);
        }
    }
#ifdef CCD_GEMM_LAB
    if ((p.m_fastest & 64) && p.colsum && t == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(p.colsum)[blockIdx.x * 8 + i] = ph[i];
#endif
}

}  // namespace ccd
