// mlp_fused.h - the whole MLP branch of a transformer block in ONE kernel (vision_transformer.py:59-65 inside
// Block.forward :107-113):
//     x_out = x_mid + rowscale * (gelu(y2 . W1^T + b1) . W2^T + b2) ;   y_next = LayerNorm(x_out) * gamma + beta
// The [rows, 4E] hidden activation never touches HBM (the teacher needs nothing else; the student additionally stores
// the bf16 pre-activation u, the one tensor its backward pass needs).  Per 131 072 rows at E = 384 the unfused pair
// (fc1 + GELU, fc2 + residual + LayerNorm) moves 1.4 GB; this kernel moves 0.6 GB (+ 0.4 GB for u).
//
// Structure ("row owners"): one workgroup = 4 waves = 128 rows, one wave per SIMD with the full 512-register file.
// A wave OWNS 32 rows for the whole kernel: their y2 values live in registers as MFMA operands (E/4 VGPRs), their
// x_out accumulators too (E/2 VGPRs); nothing of a row is ever exchanged between waves.  Only the WEIGHTS move: W1 and
// W2 (L2-resident, 2.4 MB at E = 384) stream through a ring of LDS slots by LDS-DMA, in "pieces" of 32 * E * 2 bytes that
// all four waves consume in lock step, 24 MFMAs per wave and piece (E = 384):
//     hidden chunk c (64 units):  P1(c,0) P1(c,1)  -> H^T[64 hidden][32 rows] = W1[chunk] . y2^T          (K = E,
//                                 one piece per K half, both 32-unit tiles in each: consecutive MFMAs alternate between
//                                 two accumulators - a chain on ONE accumulator with other instructions in between
//                                 pays the full write-back latency per MFMA)
//                                 GELU in registers: the accumulator layout of H^T IS the B-operand layout of the next
//                                 product once W1's rows are fed in an order that swaps bits 2 and 3 of the row index
//                                 P2(c,0) P2(c,1)  -> OUT^T[E][32 rows] += W2[:, chunk] . H                (K = 64)
// Ring protocol (NSLOT = 5 slots, every wave issues 1/4 of every piece and all waves consume every piece): step q waits
// for its own quarter of piece q (counted vmcnt: the three younger pieces stay in flight), passes ONE LDS-only barrier
// (everybody's quarter has landed AND everybody is done reading piece q-1), re-fills the slot of piece q-1 with piece
// q+4 and multiplies.  The piece sequence only depends on the position inside a row tile, so the ring runs seamlessly
// across the tiles of the persistent loop.
// Software pipeline: the pieces are consumed in the order P1(0) | P1(1) P2(0) | P1(2) P2(1) | ... | P2(last), and the GELU
// of chunk c (a 1536-entry LDS table of Phi over bf16 magnitudes, 8 VALU + 1 gather per element) is spread between the
// MFMAs of P1(c+1): with one wave per SIMD nothing else could cover it.  Fragment reads are issued by hand 6 steps ahead
// of their MFMA and retired by counted lgkmcnt waits (the compiler's own waits drain the LDS queue to zero).
//
// LDS images are the GEMM family's: 128-byte rows, 16-byte slots XOR-swizzled by f(row) on the DMA's SOURCE address
// (LDS-DMA writes lane-linearly), conflict-free ds_read_b128 fragment reads.
#pragma once

namespace ccd {

struct MlpParams {
    const bf16_t* y;        // [M, E] bf16: LayerNorm-2 output
    long ldy_in;
    const bf16_t* w1;       // fc1.weight [H, E] bf16
    long ld1;
    const float* b1;        // [H]
    const bf16_t* w2;       // fc2.weight [E, H] bf16
    long ld2;
    const float* b2;        // [E]
    const float* resid;     // x_mid [M, E] fp32
    long ldr;
    const float* rowscale;  // DropPath scale per sample, or null
    int rows_per_sample;
    float* out;             // x_out [M, E] fp32
    long ldc;
    const float* ln_gamma;  // LayerNorm that follows (norm1 of the next block / final norm)
    const float* ln_beta;
    float ln_eps;
    bf16_t* ln_y;           // [M, E] bf16
    long ld_y;
    float* ln_mean;
    float* ln_rstd;
    bf16_t* u;              // optional [M, H] bf16 pre-activation (STORE_U)
    long ldu;
    bf16_t* gact;           // optional (with u) [M, H] bf16 gelu(u): what the weight-gradient product dW2 = gb^T . gelu(u) reads.  Stored
    long ldga;              // here, the gelu'(u) product of the backward pass neither gathers Phi a second time nor writes gelu(u)
    int M, H;
    int lab;                // experiment switch (policy key "lab"): n > 0 delays odd workgroups by ~n * 8 k cycles
    // PROJ (round 5): the tail of the attention branch in front of the MLP, in the same launch -
    //     x_mid = resid + rowscale1 * (a . Wp^T + bp) ;   y2 = LayerNorm2(x_mid)   (vision_transformer.py:108-110)
    // x_mid stays in the accumulators (the second product adds onto it), y2 in the operand registers: neither is read back from
    // HBM, and a forward pass that keeps nothing for a backward pass (the teacher) does not write them either.  `y` is unused.
    const bf16_t* a;        // [M, E] bf16: attention output (heads concatenated), the projection's input
    long lda;
    const bf16_t* wp;       // proj.weight [E, E] bf16
    long ldp;
    const float* bp;        // [E]
    const float* rowscale1; // DropPath scale of the attention branch per sample, or null (rows_per_sample % 128 == 0 with either scale)
    const float* ln2_gamma; // norm2
    const float* ln2_beta;
    float* xmid;            // optional [M, E] fp32: x_mid, with y2 / mean2 / rstd2 (all four or none): what the backward pass reads
    long ldxm;
    bf16_t* y2;
    long ldy2;
    float* mean2;
    float* rstd2;
    // a second LayerNorm of the SAME output rows (other gamma / beta, same statistics): the segmentation tap that follows some blocks
    // (vision_transformer.py:245-249, norm_seg) - optional, PROJ only
    const float* tap_gamma;
    const float* tap_beta;
    bf16_t* tap_y;          // [M, E] bf16 or null
    long ld_tap;
};

constexpr int MLP_SCRATCH = 4096, MLP_THREADS = 256, MLP_BM = 128;
// ring slots: 5 (four pieces ahead) where 160 KiB allow it; 3 at E = 512 (32-KiB pieces, H = 2048 vectors)
__host__ __device__ constexpr int mlp_slots(int E) { return E <= 384 ? 5 : 3; }
__host__ __device__ constexpr int mlp_piece_bytes(int E) { return 32 * E * 2; }
__host__ __device__ inline int mlp_smem_bytes(int E, int H, bool proj = false) {
    return mlp_slots(E) * mlp_piece_bytes(E) + 4 * MLP_SCRATCH + (1536 + H + (proj ? 8 : 3) * E) * 4;    // ring, scratch, Phi table, vectors
}
// 16-byte slot swizzle of a 128-byte image row (rows taken modulo 32: a piece is a stack of 32-row blocks)
__device__ __forceinline__ int mlp_swz(int row) { return (((row & 31) >> 1) ^ ((row & 31) >> 4)) & 7; }

template <int I, int N, typename F>
__device__ __forceinline__ void mlp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        mlp_static_for<I + 1, N>(f);
    }
}
// fragment k of a piece: address register and immediate offset
template <int KTH>
struct MlpMapP1 {          // [KTH k-tiles][64 hidden rows][128 B]; MFMA k = (tile k & 1, k-step k >> 1 of this K half)
    static constexpr int reg(int k) { return (k >> 1) & 3; }
    static constexpr int off(int k) { return (k & 1) * 4096 + (k >> 3) * 8192; }
};
template <int NTH>
struct MlpMapP2 {          // [NTH tiles of 32 rows][128 B]: MFMA k = (k-step k / NTH, tile k % NTH)
    static constexpr int reg(int k) { return k / NTH; }
    static constexpr int off(int k) { return (k % NTH) * 4096; }
};
// N MFMAs of one piece: fragments requested DEPTH steps ahead, each retired by a counted wait right before its MFMA.
// Step j is [wait j][MFMA j][read j + DEPTH][filler j], and filler j issues Extra::at(j) further LDS operations by hand
// (GELU gathers): the wait of step k must leave exactly the operations issued BEHIND read k in flight, whoever issued them.
struct MlpNoExtra { static constexpr int at(int) { return 0; } };
template <int N, int DEPTH, typename Extra>
constexpr int mlp_behind(int k) {
    int n = 0;
    if (k < DEPTH) {
        n += DEPTH - 1 - k;
        for (int j = 0; j < k; ++j) n += (j + DEPTH < N ? 1 : 0) + Extra::at(j);
    } else {
        n += Extra::at(k - DEPTH);
        for (int j = k - DEPTH + 1; j < k; ++j) n += (j + DEPTH < N ? 1 : 0) + Extra::at(j);
    }
    return n;
}
template <int N, int DEPTH, typename Map, typename Extra, int NREG, typename Mma, typename Filler>
__device__ __forceinline__ void mlp_product(const unsigned (&areg)[NREG], Mma mma, Filler filler) {
    bf16x8 fr[DEPTH];
    mlp_static_for<0, DEPTH>([&](auto K) {
        constexpr int k = decltype(K)::value;
        lds_read_frag<Map::off(k)>(fr[k], areg[Map::reg(k)]);
    });
    mlp_static_for<0, N>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int behind = mlp_behind<N, DEPTH, Extra>(k);
        lds_wait_frag<(behind < 15 ? behind : 15)>(fr[k % DEPTH]);      // 4-bit counter: waiting for less is still correct
        mma(K, fr[k % DEPTH]);
        if constexpr (k + DEPTH < N) lds_read_frag<Map::off(k + DEPTH)>(fr[k % DEPTH], areg[Map::reg(k + DEPTH)]);
        filler(K);
    });
}

// GELU pairs between the MFMAs of a first-product piece: pair q (8 per piece) is issued at step q * S (two gathers) and
// finished LAG steps later
template <int KJ, int DEPTH>
struct MlpGeluSchedule {
    static constexpr int S = (KJ - DEPTH - 2) / 8 > 0 ? (KJ - DEPTH - 2) / 8 : 1, LAG = DEPTH + 1;
    static constexpr int at(int k) { return (k % S == 0 && k / S < 8) ? 2 : 0; }
};
constexpr unsigned MLP_LUT_LO = 0x3B80u, MLP_LUT_HI = 0x4180u;      // bf16 magnitudes 2^-8 .. 16: 1536 table entries

template <int E, bool STORE_U, bool PROJ = false>
__global__ __launch_bounds__(MLP_THREADS, 1) void mlp_fused_kernel(MlpParams p) {
    constexpr int KT = E / 64;             // 64-wide k-tiles of a W1 piece = DMA instructions per wave and piece
    constexpr int KJ = E / 16;             // MFMA k-steps of the first product
    constexpr int NT = E / 32;             // 32-column output tiles of a row
    constexpr int NTH = NT / 2;            // ... per W2 piece
    constexpr int KC = E / 64;             // PROJ: 64-wide k-chunks of the projection (two pieces each, as a W2 chunk)
    constexpr int NPP = PROJ ? 2 * KC : 0; // PROJ: pieces of the projection in front of a row tile's stream
    constexpr int PIECE = mlp_piece_bytes(E);
    constexpr int MLP_NSLOT = mlp_slots(E);
    constexpr int AHEAD = MLP_NSLOT - 1;   // pieces issued ahead of the one being consumed
    constexpr int DEPTH = 6;               // fragment reads in flight ahead of their MFMA
    static_assert(E % 128 == 0 && AHEAD >= 2, "ring bookkeeping; a W1 piece is one K half in whole 64-wide k-tiles");
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, hf = lane >> 5, lq = lane & 31;
    const int w = uniform_i32(t >> 6);     // wave index as a scalar: everything derived from it stays in SGPRs
    char* scratch = smem + MLP_NSLOT * PIECE + w * MLP_SCRATCH;
    float* lut = reinterpret_cast<float*>(smem + MLP_NSLOT * PIECE + 4 * MLP_SCRATCH);
    float* vb1 = lut + (MLP_LUT_HI - MLP_LUT_LO);
    float* vb2 = vb1 + p.H;
    float* vga = vb2 + E;
    float* vbe = vga + E;
    float* vbp = vbe + E;                  // PROJ: projection bias, norm2 gamma / beta
    float* vga2 = vbp + E;
    float* vbe2 = vga2 + E;
    float* vgat = vbe2 + E;                // PROJ: the tap's gamma / beta
    float* vbet = vgat + E;
    for (int i = t; i < p.H; i += MLP_THREADS) vb1[i] = p.b1[i];
    for (int i = t; i < E; i += MLP_THREADS) { vb2[i] = p.b2[i]; vga[i] = p.ln_gamma[i]; vbe[i] = p.ln_beta[i]; }
    if constexpr (PROJ) {
        for (int i = t; i < E; i += MLP_THREADS) { vbp[i] = p.bp[i]; vga2[i] = p.ln2_gamma[i]; vbe2[i] = p.ln2_beta[i]; }
        if (p.tap_y)
            for (int i = t; i < E; i += MLP_THREADS) { vgat[i] = p.tap_gamma[i]; vbet[i] = p.tap_beta[i]; }
    }
    for (unsigned i = t; i < MLP_LUT_HI - MLP_LUT_LO; i += MLP_THREADS) lut[i] = gelu_terms(bf2f((bf16_t)(MLP_LUT_LO + i))).cdf;
    __syncthreads();                       // (plain loads only so far: nothing in flight that a drain would hurt)

    const int NC = p.H / 64, NP = NPP + 4 * NC;  // hidden chunks, pieces per row tile
    const int tiles = (p.M + MLP_BM - 1) / MLP_BM, G = gridDim.x;
#ifdef CCD_MLP_LAB      // lab build only: cycle totals of wave 0 per phase -> first 64 bytes per workgroup of ln_mean (destroyed)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define MLP_STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tprev; tprev = tn_; }
#else
#define MLP_STAMP(i)
#endif

    // ---- DMA: a piece is 4*KT wave instructions of 1 KiB (8 image rows x 128 B).  Wave w moves the instructions whose
    // 8-row block index is w modulo 4, so ONE per-lane offset per weight matrix serves all of its instructions:
    //   W1 piece [KT/2 k-tiles of one K half][64 hidden rows][128 B]: instruction i = rows 32 (i & 1) + 8w .. + 7 of k-tile i >> 1
    //   W2 piece [E/2 output rows][128 B = 64 hidden units]: instruction i = rows 32 i + 8w .. + 7
    //   Wp piece (PROJ) = a W2 piece of proj.weight: [E/2 output rows][128 B = 64 input columns]
    const int dr = lane >> 3, dp = lane & 7;
    const int drow = 8 * w + dr;
    // per-lane byte offset of a request = drow2 * (row stride of the matrix) + swz16: one multiply-add where it is needed
    // (two precomputed offsets selected by the piece type became a scratch array)
    const unsigned drow2_ = (unsigned)(2 * drow), swz16_ = (unsigned)((dp ^ mlp_swz(drow)) * 16);
    int slot_i = 0, slot_c = 0, pos_i = 0; // ring slots of the next request / consumption, stream position of the next request
    // stream of a row tile (NP pieces): [PROJ: Pp(0)a Pp(0)b ... Pp(KC-1)a Pp(KC-1)b |] P1(0)a P1(0)b | P1(1)a P1(1)b P2(0)a P2(0)b | ... | P2(NC-1)a P2(NC-1)b
    // A request is PREPARED once (scalar state: where the piece starts, its two strides, its ring slot) and its KT
    // instructions are then issued one at a time between the MFMAs of the piece being consumed: an LDS-DMA instruction keeps
    // the issuing wave busy for tens of cycles, and 6 of them back to back (with their address arithmetic) measured a quarter
    // of the kernel.  Instruction i moves the 8 rows at  base + (i & 1) * step_a + (i >> 1) * step_b  to  slot + 4096 * i.
    const char* req_base = nullptr;
    long req_step_a = 0, req_step_b = 0;
    unsigned req_lane = 0;
    char* req_lds = nullptr;
    auto issue_prepare = [&]() __attribute__((always_inline)) {
        int is_p2 = 0, chunk = 0, half = 0, pi = pos_i;
        // (opaque: with the stream positions of a ring_seek known at compile time the optimiser otherwise precomputes every
        // request's per-lane address outside the tile loop - 2 x 24 64-bit values that it then keeps in scratch)
        const unsigned drow2 = (unsigned)opaque_vgpr((int)drow2_), swz16 = (unsigned)opaque_vgpr((int)swz16_);
        bool is_pp = false;
        if constexpr (PROJ) {
            if (pi < NPP) is_pp = true;
            else pi -= NPP;
        }
        if (is_pp) { chunk = pi >> 1; half = pi & 1; }
        else if (pi < 2) { is_p2 = 0; chunk = 0; half = pi; }
        else {
            const int q = pi - 2, grp = q >> 2, r = q & 3;
            if (grp < NC - 1) { is_p2 = r >> 1; chunk = is_p2 ? grp : grp + 1; half = r & 1; }
            else { is_p2 = 1; chunk = NC - 1; half = r; }
        }
        if (is_pp) {       // rows half * E/2 + 32 i + .., input columns 64 chunk ..
            req_base = reinterpret_cast<const char*>(p.wp) + ((long)(half * (E / 2)) * p.ldp + 64 * chunk) * 2;
            req_step_a = 64 * p.ldp;
            req_step_b = 128 * p.ldp;
            req_lane = drow2 * (unsigned)p.ldp + swz16;
        } else if (!is_p2) {      // rows 64 chunk + 32 (i & 1) + .., K half `half`, k-tile i >> 1
            req_base = reinterpret_cast<const char*>(p.w1) + ((long)(64 * chunk) * p.ld1 + half * (E / 2)) * 2;
            req_step_a = 64 * p.ld1;
            req_step_b = 128;
            req_lane = drow2 * (unsigned)p.ld1 + swz16;
        } else {           // rows half * E/2 + 32 i + .., hidden units 64 chunk ..
            req_base = reinterpret_cast<const char*>(p.w2) + ((long)(half * (E / 2)) * p.ld2 + 64 * chunk) * 2;
            req_step_a = 64 * p.ld2;
            req_step_b = 128 * p.ld2;
            req_lane = drow2 * (unsigned)p.ld2 + swz16;
        }
        req_lds = smem + slot_i * PIECE + w * 1024;
        slot_i = slot_i + 1 == MLP_NSLOT ? 0 : slot_i + 1;
        pos_i = pos_i + 1 == NP ? 0 : pos_i + 1;
    };
    auto issue_one = [&](int i) __attribute__((always_inline)) {
        glds16(req_base + ((i & 1) * req_step_a + (i >> 1) * req_step_b) + req_lane, req_lds + 4096 * i);
    };
    // step of the ring: my quarter of the next piece has landed (three younger pieces stay in flight), everybody's has
    // and everybody is done with the previous piece (barrier), whose slot is re-filled 4 pieces ahead - by the instructions
    // the caller spreads over its product (dma_slot below).  Requests run past the last tile (the weights are the same for
    // every tile; the surplus is drained at the end).
    const unsigned smem_addr = lds_addr_of(smem);
    auto acquire = [&]() __attribute__((always_inline)) -> unsigned {
        glds_wait<(AHEAD - 1) * KT>();
        lds_barrier();
        MLP_STAMP(1)                         // lab: wait for the DMA + barrier
        issue_prepare();
        MLP_STAMP(2)                         // lab: request bookkeeping
        const unsigned sb = smem_addr + (unsigned)(slot_c * PIECE);
        slot_c = slot_c + 1 == MLP_NSLOT ? 0 : slot_c + 1;
        return sb;
    };
    // DMA instruction i of the prepared request goes out behind MFMA step 1 + i * (N / KT) of an N-step product
    auto dma_slot = [&](auto K, auto NSTEPS) {
        constexpr int k = decltype(K)::value, stride = decltype(NSTEPS)::value / KT;
        if constexpr (k % stride == 1 && k / stride < KT) issue_one(k / stride);
    };
    auto ring_fill = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) {
            issue_prepare();
#pragma unroll
            for (int i = 0; i < KT; ++i) issue_one(i);
        }
    };
    ring_fill();

    // ---- fragment read offsets inside a piece (one register per k-step; tiles / k-tiles are immediate offsets)
    const int prow = (lq & 19) | ((lq & 4) << 1) | ((lq & 8) >> 1);      // bits 2 and 3 of the row index swapped
    // offset of k-step kk = off[0] ^ (kk << 5): 2 kk sits in bits 1-2 of the 16-byte slot index, the swizzle is an XOR on the same
    // bits (gemm256.h's form) - ONE register per image kind lives across the main loop instead of four
    const unsigned off1_0 = (unsigned)(prow * 128 + ((hf ^ mlp_swz(prow)) * 16));
    const unsigned off2_0 = (unsigned)(lq * 128 + ((hf ^ mlp_swz(lq)) * 16));
    const float inv_e = 1.0f / (float)E;
    // global traffic of the row tiles goes through buffer descriptors: one per-lane offset register per tensor, the tile /
    // column part of every address in SGPRs, and rows beyond M cost no predicate (loads return 0, stores are dropped)
    // (the descriptors of the tensors that only the row passes touch are built where they are used, on a size the optimiser cannot hoist
    // - a dozen of them held in scalar registers across the product loops had the allocator spill 122 scalars into vector lanes and read
    // 16 of them back per weight piece, each a slot of the issue stream the MFMAs share)
#define MLP_RS_Y (PROJ ? make_rsrc(p.a, opaque_u32((unsigned)((((long)p.M - 1) * p.lda + E) * 2))) \
                       : make_rsrc(p.y, opaque_u32((unsigned)((((long)p.M - 1) * p.ldy_in + E) * 2))))
#define MLP_RS_X make_rsrc(p.resid, opaque_u32((unsigned)((((long)p.M - 1) * p.ldr + E) * 4)))
#define MLP_RS_O make_rsrc(p.out, opaque_u32((unsigned)((((long)p.M - 1) * p.ldc + E) * 4)))
#define MLP_RS_N make_rsrc(p.ln_y, opaque_u32((unsigned)((((long)p.M - 1) * p.ld_y + E) * 2)))
    const long ld_yin = PROJ ? p.lda : p.ldy_in;
    const buf_rsrc rs_u = make_rsrc(STORE_U ? p.u : nullptr, STORE_U ? (unsigned)((((long)p.M - 1) * p.ldu + p.H) * 2) : 0u);
    const bool store_g = STORE_U && p.gact != nullptr;
    const buf_rsrc rs_ga = make_rsrc(store_g ? p.gact : nullptr, store_g ? (unsigned)((((long)p.M - 1) * p.ldga + p.H) * 2) : 0u);
    constexpr bool keep_mid = PROJ && STORE_U;           // x_mid, y2 and their statistics are written with u (the host checks: all or none)
    // per-lane offsets of the row-tile traffic are recomputed from the lane id where they are used (LaneOff below): kept in
    // registers across the main loop they were the first values the allocator spilled, and a scratch reload in front of
    // every store (s_waitcnt vmcnt(0)!) serialised the whole epilogue
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
    if (p.lab > 0 && (blockIdx.x & 1)) wave_sleep(p.lab);
    for (int tile = blockIdx.x; tile < tiles; tile += G) {
        const int m0 = tile * MLP_BM, r0 = m0 + 32 * w;
        const int row = r0 + lq, grow = row < p.M ? row : p.M - 1;
        // DropPath: out = x + sc * (h . W2^T + b2), sc per sample (0 for a dropped one).  A tile of dropped samples only
        // (rows_per_sample a multiple of the tile height) skips the products altogether.
        // A tile inside ONE sample has one scale: read through the scalar cache (lgkmcnt).  A vector load here - even on a path
        // that is not taken - puts `s_waitcnt vmcnt(0)` at the top of every tile: the weight ring's requests and the previous
        // tile's stores drained before the tile's own rows are even requested.  A tile that spans samples reads its per-row
        // scale where it is used, in the epilogue.
        float sc_tile = 1.0f, sc1_tile = 1.0f;
        bool tile_dead = false;
        const bool one_sample = p.rows_per_sample % MLP_BM == 0;
        if (p.rowscale && one_sample) {
            sc_tile = scalar_load_f32(uniform_ptr(p.rowscale + uniform_i32(m0 / p.rows_per_sample)));
            tile_dead = sc_tile == 0.0f;
        }
        if constexpr (PROJ) {
            // (the host guarantees one_sample whenever PROJ comes with a scale.)  A dropped branch is NOT skipped here: its products run
            // and are discarded - scale 0 for the projection; for the MLP branch x_mid is read back at the end - so that the common
            // path has no branches around the ring (with ring seeks over the dropped pieces the register allocator moved a third
            // of the accumulators through scratch at every join: 227 spilled registers).  Dropped tiles are ~5 % of the student's.
            if (p.rowscale1) sc1_tile = scalar_load_f32(uniform_ptr(p.rowscale1 + uniform_i32(m0 / p.rows_per_sample)));
        }
        // x_out = x, y_next = LayerNorm(x) for this wave's 32 rows: half a wave per row, row sums by shuffles (as gemm_row384.h's epilogue)
        auto dead_rows = [&](const buf_rsrc& rs_dst, long ld_dst, const float* ga_, const float* be_, bf16_t* yp, long ldyp,
                             float* meanp, float* rstdp) __attribute__((always_inline)) {
            constexpr int C3 = (E / 4 + 31) / 32;              // 16-byte chunks of a row per lane
            const buf_rsrc rs_x = MLP_RS_X;
#pragma unroll 1
            for (int it = 0; it < 16; ++it) {
                const unsigned rr = (unsigned)(r0 + 2 * it + hf);
                f32x4v o[C3];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int c3 = 0; c3 < C3; ++c3) {
                    const int chunk = lq + 32 * c3;
                    o[c3] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    if (chunk < E / 4) o[c3] = __builtin_bit_cast(f32x4v, buf_load16(rs_x, (unsigned)(chunk * 16), rr * (unsigned)(p.ldr * 4)));
                    s1 += (o[c3].x + o[c3].y) + (o[c3].z + o[c3].w);
                    s2 += (o[c3].x * o[c3].x + o[c3].y * o[c3].y) + (o[c3].z * o[c3].z + o[c3].w * o[c3].w);
                }
#pragma unroll
                for (int msk = 16; msk >= 1; msk >>= 1) { s1 += shfl_xor(s1, msk); s2 += shfl_xor(s2, msk); }
                const float mean = s1 * inv_e;
                float var = s2 * inv_e - mean * mean;
                var = var > 0.f ? var : 0.f;
                const float rstd = 1.0f / sqrtf(var + p.ln_eps);
#pragma unroll
                for (int c3 = 0; c3 < C3; ++c3) {
                    const int chunk = lq + 32 * c3;
                    if (chunk < E / 4) {
                        buf_store16(rs_dst, (unsigned)(chunk * 16), rr * (unsigned)(ld_dst * 4), __builtin_bit_cast(u32x4, o[c3]));
                        const f32x4v ga = *reinterpret_cast<const f32x4v*>(ga_ + 4 * chunk), be = *reinterpret_cast<const f32x4v*>(be_ + 4 * chunk);
                        u32x2 pk;
                        pk.x = pack_bf2((o[c3].x - mean) * rstd * ga.x + be.x, (o[c3].y - mean) * rstd * ga.y + be.y);
                        pk.y = pack_bf2((o[c3].z - mean) * rstd * ga.z + be.z, (o[c3].w - mean) * rstd * ga.w + be.w);
                        if (rr < (unsigned)p.M) *reinterpret_cast<u32x2*>(yp + (long)rr * ldyp + 4 * chunk) = pk;
                    }
                }
                if (lq == 0 && rr < (unsigned)p.M) { meanp[rr] = mean; rstdp[rr] = rstd; }
            }
        };
        // the backward pass multiplies a zero gradient by gelu'(u) for the rows of a dropped MLP branch: u must be finite
        auto zero_u = [&]() __attribute__((always_inline)) {
            if constexpr (STORE_U) {
                const unsigned lo_u = LaneOff(t).rows8(p.ldu, 2);
#pragma unroll 1
                for (int cc = 0; cc < p.H / 64; ++cc)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        stream_store16<NT_MLP_U>(rs_u, lo_u, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldu * 2) + 128 * cc, u32x4{0u, 0u, 0u, 0u});
                if (store_g) {
                    const unsigned lo_g = LaneOff(t).rows8(p.ldga, 2);
#pragma unroll 1
                    for (int cc = 0; cc < p.H / 64; ++cc)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            buf_store16(rs_ga, lo_g, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldga * 2) + 128 * cc, u32x4{0u, 0u, 0u, 0u});
                }
            }
        };
        if (!PROJ && tile_dead) {
            dead_rows(MLP_RS_O, p.ldc, vga, vbe, p.ln_y, p.ld_y, p.ln_mean, p.ln_rstd);
            zero_u();
            continue;
        }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        bf16x8 yf[KJ];                     // this lane's row of y2 (PROJ: first of the attention output) as B operands: k = 16 j + 8 hf .. + 7
        auto load_rows = [&]() __attribute__((always_inline)) {
            const unsigned so = (unsigned)r0 * (unsigned)(ld_yin * 2), lo_y = LaneOff(t).frag(ld_yin, 2, 8);
            const buf_rsrc rs_y = MLP_RS_Y;
#pragma unroll
            for (int j = 0; j < KJ; ++j) yf[j] = __builtin_bit_cast(bf16x8, stream_load16<NT_MLP_Y>(rs_y, lo_y, so + 32 * j));
        };
        // ---- row passes over the accumulators (rows are complete inside their two lanes: lane, lane ^ 32).
        // Pass A: v = x + (acc + bias) * sc (the residual rows stream in two tiles ahead of their use), v = acc * sc or v = x, written
        // back into the accumulators, LayerNorm statistics on the way.
        float mean = 0.f, rstd = 0.f;
        auto pass_a = [&](auto MODE, const buf_rsrc& rs_src, long ld_src, const float* vbias, float sc) __attribute__((always_inline)) {
            constexpr int mode = decltype(MODE)::value;      // 0: v = acc * sc;  1: v = x + (acc + bias) * sc;  2: v = x (read past the L1)
            float s1 = 0.f, s2 = 0.f;
            if constexpr (mode != 0) {
                const unsigned so = (unsigned)r0 * (unsigned)(ld_src * 4), lo_x = LaneOff(t).frag(ld_src, 4, 4);
                u32x4 xb[3][4];
                auto load_x = [&](int nt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (mode == 2) xb[nt % 3][g] = buf_load16_coherent(rs_src, lo_x, so + (32 * nt + 8 * g) * 4);
                        else xb[nt % 3][g] = stream_load16<NT_MLP_X>(rs_src, lo_x, so + (32 * nt + 8 * g) * 4);
                    }
                };
                load_x(0);
                if (NT > 1) load_x(1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (nt + 2 < NT) load_x(nt + 2);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4v x = __builtin_bit_cast(f32x4v, xb[nt % 3][g]);
                        const f32x4v b = *reinterpret_cast<const f32x4v*>(vbias + 32 * nt + 8 * g + 4 * hf);
                        const float xx[4] = {x.x, x.y, x.z, x.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = mode == 2 ? xx[e] : xx[e] + (acc[nt][4 * g + e] + bb[e]) * sc;
                            acc[nt][4 * g + e] = v;
                            s1 += v;
                            s2 = fmaf(v, v, s2);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[nt][r] * sc;
                        acc[nt][r] = v;
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
            }
            s1 += shfl_xor(s1, 32);
            s2 += shfl_xor(s2, 32);
            mean = s1 * inv_e;
            float var = s2 * inv_e - mean * mean;
            var = var > 0.f ? var : 0.f;
            rstd = 1.0f / sqrtf(var + p.ln_eps);
        };
        // Pass B, per pair of 32-column tiles: the rows (fp32, one tile at a time) and their LayerNorm (bf16, both tiles) leave through
        // the scratch image as 128-byte row segments (STORE), and / or the LayerNorm becomes the operand registers yf (TO_YF: the two
        // half-waves exchange quads - v_permlane32_swap - so that a lane holds the 8 consecutive columns 16 j + 8 hf .. of k-step j)
        auto pass_b = [&](auto STORE, auto TO_YF, const buf_rsrc& rs_of, long ldo, const buf_rsrc& rs_nf, long ldn, const float* ga_,
                          const float* be_, float* meanp, float* rstdp, bool tap = false) __attribute__((always_inline)) {
            constexpr bool store = decltype(STORE)::value, to_yf = decltype(TO_YF)::value;
            const buf_rsrc rs_tap = make_rsrc(tap ? p.tap_y : nullptr, tap ? (unsigned)((((long)p.M - 1) * p.ld_tap + E) * 2) : 0u);
            if (store && hf == 0 && row < p.M) { meanp[row] = mean; rstdp[row] = rstd; }
            const LaneOff lo(t);
            const unsigned lo_o = lo.rows8(ldo, 4), lo_n = lo.rows8(ldn, 2);
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                u32x2 ypk[2][4];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int nt = 2 * np + tt;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float v[4] = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                        if constexpr (store) {
                            const f32x4v o = {v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4v*>(scratch + lo.scr_wr(2 * g + lo.hf)) = o;
                        }
                        const int n = 32 * nt + 8 * g + 4 * hf;
                        const f32x4v ga = *reinterpret_cast<const f32x4v*>(ga_ + n), be = *reinterpret_cast<const f32x4v*>(be_ + n);
                        ypk[tt][g].x = pack_bf2((v[0] - mean) * rstd * ga.x + be.x, (v[1] - mean) * rstd * ga.y + be.y);
                        ypk[tt][g].y = pack_bf2((v[2] - mean) * rstd * ga.z + be.z, (v[3] - mean) * rstd * ga.w + be.w);
                    }
                    if constexpr (store) {
                        wave_lds_fence();
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                            stream_store16<NT_MLP_OUT>(rs_of, lo_o, (unsigned)(r0 + 8 * i) * (unsigned)(ldo * 4) + 128 * nt, v);
                        }
                        wave_lds_fence();
                    }
                }
                if constexpr (store) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<u32x2*>(scratch + lo.scr_wr(4 * tt + g) + 8 * lo.hf) = ypk[tt][g];
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                        buf_store16(rs_nf, lo_n, (unsigned)(r0 + 8 * i) * (unsigned)(ldn * 2) + 128 * np, v);
                    }
                    wave_lds_fence();
                    if (tap) {             // the same rows under the tap's gamma / beta (same mean / rstd), same way out
                        const unsigned lo_t = lo.rows8(p.ld_tap, 2);
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int nt = 2 * np + tt, n = 32 * nt + 8 * g + 4 * hf;
                                const f32x4v ga = *reinterpret_cast<const f32x4v*>(vgat + n), be = *reinterpret_cast<const f32x4v*>(vbet + n);
                                u32x2 tp;
                                tp.x = pack_bf2((acc[nt][4 * g] - mean) * rstd * ga.x + be.x, (acc[nt][4 * g + 1] - mean) * rstd * ga.y + be.y);
                                tp.y = pack_bf2((acc[nt][4 * g + 2] - mean) * rstd * ga.z + be.z, (acc[nt][4 * g + 3] - mean) * rstd * ga.w + be.w);
                                *reinterpret_cast<u32x2*>(scratch + lo.scr_wr(4 * tt + g) + 8 * lo.hf) = tp;
                            }
                        wave_lds_fence();
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                            buf_store16(rs_tap, lo_t, (unsigned)(r0 + 8 * i) * (unsigned)(p.ld_tap * 2) + 128 * np, v);
                        }
                        wave_lds_fence();
                    }
                }
                if constexpr (to_yf) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            // own quads: columns 16 j + 4 hf .. + 3 (a) and 16 j + 8 + 4 hf .. + 3 (b) of k-step j = 2 nt + s
                            unsigned ax = ypk[tt][2 * s].x, ay = ypk[tt][2 * s].y, bx = ypk[tt][2 * s + 1].x, by = ypk[tt][2 * s + 1].y;
                            lane32_swap(ax, bx);
                            lane32_swap(ay, by);
                            const u32x4 f = {ax, ay, bx, by};
                            yf[2 * (2 * np + tt) + s] = __builtin_bit_cast(bf16x8, f);
                        }
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using Yes = std::true_type;
        using No = std::false_type;
        // acc[half hh] += (piece of W2 / Wp: E/2 output rows x 64 k) . B, B = four k-steps of packed operands
        auto p2_piece = [&](auto HH, const bf16x8& b0, const bf16x8& b1, const bf16x8& b2, const bf16x8& b3) __attribute__((always_inline)) {
            constexpr int hh = decltype(HH)::value;
            const unsigned sb = acquire();
            const unsigned a0_ = sb + off2_0, areg[4] = {a0_, a0_ ^ 32u, a0_ ^ 64u, a0_ ^ 96u};
            mlp_product<4 * NTH, DEPTH, MlpMapP2<NTH>, MlpNoExtra>(
                areg,
                [&](auto K, const bf16x8& a) {
                    constexpr int k = decltype(K)::value, ks = k / NTH;
                    const bf16x8& b = ks == 0 ? b0 : ks == 1 ? b1 : ks == 2 ? b2 : b3;
                    acc[NTH * hh + k % NTH] = mfma_32x32x16_bf16(a, b, acc[NTH * hh + k % NTH]);
                },
                [&](auto K) { dma_slot(K, std::integral_constant<int, 4 * NTH>{}); });
        };
        float sc_fin = 1.0f;                // what the accumulators are multiplied by at the end (PROJ)
        if constexpr (PROJ) {
            // ---- the attention branch's tail: acc = a . Wp^T, then x_mid = x + (acc + bp) * sc1 and its LayerNorm
            load_rows();
            MLP_STAMP(0)
            mlp_static_for<0, KC>([&](auto KCI) {
                constexpr int kc = decltype(KCI)::value;
                p2_piece(I0{}, yf[4 * kc], yf[4 * kc + 1], yf[4 * kc + 2], yf[4 * kc + 3]);
                p2_piece(I1{}, yf[4 * kc], yf[4 * kc + 1], yf[4 * kc + 2], yf[4 * kc + 3]);
            });
            MLP_STAMP(6)
            pass_a(std::integral_constant<int, 1>{}, MLP_RS_X, p.ldr, vbp, sc1_tile);
            // the second product accumulates ON x_mid: acc = x_mid / sc2 + b2, x_out = acc * sc2 (a dropped MLP branch, sc2 = 0: the
            // products run on whatever and x_mid is read back at the end)
            sc_fin = tile_dead ? 1.0f : sc_tile;
            const float inv2 = 1.0f / sc_fin;
            if constexpr (keep_mid) {
                const buf_rsrc rs_xm = make_rsrc(p.xmid, (unsigned)((((long)p.M - 1) * p.ldxm + E) * 4));
                const buf_rsrc rs_y2 = make_rsrc(p.y2, (unsigned)((((long)p.M - 1) * p.ldy2 + E) * 2));
                pass_b(Yes{}, Yes{}, rs_xm, p.ldxm, rs_y2, p.ldy2, vga2, vbe2, p.mean2, p.rstd2);
            } else {
                pass_b(No{}, Yes{}, MLP_RS_O, p.ldc, MLP_RS_N, p.ld_y, vga2, vbe2, p.mean2, p.rstd2);
            }
            MLP_STAMP(7)
            // (one 32-column tile at a time: left to itself the scheduler reads all 192 accumulators into registers before it writes
            // the first one back, on top of the 96 operand registers just built)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4v b = *reinterpret_cast<const f32x4v*>(vb2 + 32 * nt + 8 * g + 4 * hf);
                    acc[nt][4 * g] = fmaf(acc[nt][4 * g], inv2, b.x);
                    acc[nt][4 * g + 1] = fmaf(acc[nt][4 * g + 1], inv2, b.y);
                    acc[nt][4 * g + 2] = fmaf(acc[nt][4 * g + 2], inv2, b.z);
                    acc[nt][4 * g + 3] = fmaf(acc[nt][4 * g + 3], inv2, b.w);
                }
                CCD_SCHED_FENCE();
            }
        } else {
            load_rows();
        }
        {
            f32x16 h[2];                   // H^T of the chunk being produced: 2 tiles of [32 hidden][32 rows]
            u32x4 hbw[4];                  // gelu(H) of the chunk being consumed, as packed bf16 B operands (k-step s = hbw[s])
            const unsigned lut_addr = lds_addr_of(lut) - 4u * MLP_LUT_LO;       // byte address of entry "magnitude 0"
            auto p1_piece = [&](auto KH, int chunk, auto extra, auto filler) {   // h += W1[chunk][:, K half kh] . y2[K half kh]^T
                constexpr int kh = decltype(KH)::value;
                using Extra = decltype(extra);
                const unsigned sb = acquire();
                const unsigned a0_ = sb + off1_0, areg[4] = {a0_, a0_ ^ 32u, a0_ ^ 64u, a0_ ^ 96u};
                if constexpr (kh == 0) {
                    // the accumulators start at the bias: register r of tile tt is hidden unit 32 tt + 16 (r >> 3) + 8 hf +
                    // (r & 7) for every row (column of H^T).  Read here, with the LDS queue empty, not between the MFMAs.
                    const float* bp = vb1 + 64 * chunk + 8 * hf;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4v b = *reinterpret_cast<const f32x4v*>(bp + 32 * tt + 16 * (q >> 1) + 4 * (q & 1));
                            h[tt][4 * q] = b.x; h[tt][4 * q + 1] = b.y; h[tt][4 * q + 2] = b.z; h[tt][4 * q + 3] = b.w;
                        }
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
            if constexpr (!PROJ) { MLP_STAMP(0) }
            p1_piece(I0{}, 0, MlpNoExtra{}, [](auto) {});
            MLP_STAMP(3)
            p1_piece(I1{}, 0, MlpNoExtra{}, [](auto) {});
            MLP_STAMP(3)
#pragma unroll 1
            for (int c = 0; c < NC; ++c) {
                // GELU of the bf16-rounded pre-activation (what the backward pass will see) of chunk c, two elements at a
                // time ("pairs", 8 per tile): registers 8 s + (0 .. 7) of tile tt are hidden units 64 c + 32 tt + 16 s + 8 hf
                // + (0 .. 7), and the 16 values of a (tt, s) pair ARE the B operand of k-step 2 tt + s of the second product.
                //   gelu(u) = |u| Phi(|u|) + min(u, 0),  Phi(|u|) gathered from the LDS table by the bf16 magnitude.
                // A pair is ISSUED (pack, index, two hand-issued gathers) at one MFMA step and FINISHED DEPTH + 1 steps later,
                // when the counted wait of a fragment read that was issued behind its gathers has passed (LDS returns in
                // order) - no wait of its own, nothing the compiler would turn into lgkmcnt(0) between the MFMAs.
                const f32x16 hv[2] = {h[0], h[1]};
                unsigned upk[8];               // (up to LAG / S + 1 pairs are in flight; slots are static indices)
                float f0[8], f1[8];
                u32x4 uw;
                auto pair_issue = [&](auto PI) {
                    constexpr int pi = decltype(PI)::value, k4 = pi >> 2, e = pi & 3, r = 8 * (k4 & 1) + 2 * e, sl = pi & 7;
                    upk[sl] = pack_bf2(hv[k4 >> 1][r], hv[k4 >> 1][r + 1]);
                    unsigned m0_ = upk[sl] & 0x7fffu, m1_ = (upk[sl] >> 16) & 0x7fffu;
                    m0_ = m0_ < MLP_LUT_LO ? MLP_LUT_LO : (m0_ > MLP_LUT_HI - 1u ? MLP_LUT_HI - 1u : m0_);
                    m1_ = m1_ < MLP_LUT_LO ? MLP_LUT_LO : (m1_ > MLP_LUT_HI - 1u ? MLP_LUT_HI - 1u : m1_);
                    lds_gather_f32(f0[sl], lut_addr + 4u * m0_);
                    lds_gather_f32(f1[sl], lut_addr + 4u * m1_);
                };
                auto pair_finish = [&](auto PI) {
                    constexpr int pi = decltype(PI)::value, k4 = pi >> 2, e = pi & 3, sl = pi & 7;
                    lds_landed(f0[sl], f1[sl]);
                    const float u0 = bf_lo(upk[sl]), u1 = bf_hi(upk[sl]);
                    const float g0 = fmaf(fabsf(u0), f0[sl], fminf(u0, 0.f));
                    const float g1 = fmaf(fabsf(u1), f1[sl], fminf(u1, 0.f));
                    hbw[k4][e] = pack_bf2(g0, g1);
                    if (STORE_U) {
                        uw[e] = upk[sl];
                        if (e == 3) {      // [32 rows][64 hidden] bf16 image in the wave's scratch
                            const LaneOff lo(t);
                            *reinterpret_cast<u32x4*>(scratch + lo.scr_wr(2 * k4 + lo.hf)) = uw;
                        }
                    }
                };
                if (c + 1 < NC) {
                    // chunk c+1's first product with chunk c's GELU between its MFMAs: pair q of a piece (8 per piece) is
                    // issued at step q * S and finished at step q * S + DEPTH + 1 when that step exists, else after the piece
                    auto piece_with_gelu = [&](auto KH) {
                        constexpr int kh = decltype(KH)::value;
                        p1_piece(KH, c + 1, MlpGeluSchedule<KJ, DEPTH>{}, [&](auto K) {
                            constexpr int k = decltype(K)::value;
                            using Sch = MlpGeluSchedule<KJ, DEPTH>;
                            if constexpr (k >= Sch::LAG && (k - Sch::LAG) % Sch::S == 0 && (k - Sch::LAG) / Sch::S < 8)
                                pair_finish(std::integral_constant<int, 8 * kh + (k - Sch::LAG) / Sch::S>{});
                            if constexpr (Sch::at(k) != 0) pair_issue(std::integral_constant<int, 8 * kh + k / Sch::S>{});
                        });
                        constexpr int first_late = (KJ - 1 - MlpGeluSchedule<KJ, DEPTH>::LAG) / MlpGeluSchedule<KJ, DEPTH>::S + 1;
                        if constexpr (first_late < 8) {      // pairs whose finishing step does not exist
                            lds_drain();
                            mlp_static_for<8 * kh + (first_late < 0 ? 0 : first_late), 8 * kh + 8>(pair_finish);
                        }
                    };
                    piece_with_gelu(I0{});
                    MLP_STAMP(4)
                    piece_with_gelu(I1{});
                    MLP_STAMP(4)
                } else {
                    mlp_static_for<0, 4>([&](auto Gq) {
                        constexpr int gq = decltype(Gq)::value;
                        mlp_static_for<4 * gq, 4 * gq + 4>(pair_issue);
                        lds_drain();
                        mlp_static_for<4 * gq, 4 * gq + 4>(pair_finish);
                    });
                }
                if (STORE_U) {             // ... leaves as 128-byte row segments
                    wave_lds_fence();
                    const LaneOff lo(t);
                    const unsigned lo_u = lo.rows8(p.ldu, 2);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                        stream_store16<NT_MLP_U>(rs_u, lo_u, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldu * 2) + 128 * c, v);
                    }
                    wave_lds_fence();
                    if (store_g) {         // gelu(u): the packed B operands of the second product ARE the image's 16-byte slots
                        const unsigned lo_g = lo.rows8(p.ldga, 2);
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) *reinterpret_cast<u32x4*>(scratch + lo.scr_wr(2 * k4 + lo.hf)) = hbw[k4];
                        wave_lds_fence();
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                            buf_store16(rs_ga, lo_g, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldga * 2) + 128 * c, v);
                        }
                        wave_lds_fence();
                    }
                }
                MLP_STAMP(5)
                p2_piece(I0{}, __builtin_bit_cast(bf16x8, hbw[0]), __builtin_bit_cast(bf16x8, hbw[1]), __builtin_bit_cast(bf16x8, hbw[2]),
                         __builtin_bit_cast(bf16x8, hbw[3]));
                MLP_STAMP(6)
                p2_piece(I1{}, __builtin_bit_cast(bf16x8, hbw[0]), __builtin_bit_cast(bf16x8, hbw[1]), __builtin_bit_cast(bf16x8, hbw[2]),
                         __builtin_bit_cast(bf16x8, hbw[3]));
                MLP_STAMP(6)
            }
        }
        // ---- epilogue: x_out into the accumulators + statistics, then x_out and y_next leave
        if constexpr (PROJ) {
            if (tile_dead) {                // (rare) x_out = x_mid exactly: this wave's own rows, written ~100 us ago, read past the L1
                if constexpr (keep_mid) {
                    const buf_rsrc rs_xm = make_rsrc(p.xmid, (unsigned)((((long)p.M - 1) * p.ldxm + E) * 4));
                    pass_a(std::integral_constant<int, 2>{}, rs_xm, p.ldxm, vb2, 1.0f);
                }
            } else {
                pass_a(std::integral_constant<int, 0>{}, MLP_RS_X, p.ldr, vb2, sc_fin);
            }
        } else {
            const float sc = (p.rowscale && !one_sample) ? p.rowscale[grow / p.rows_per_sample] : sc_tile;
            pass_a(std::integral_constant<int, 1>{}, MLP_RS_X, p.ldr, vb2, sc);
        }
        pass_b(Yes{}, No{}, MLP_RS_O, p.ldc, MLP_RS_N, p.ld_y, vga, vbe, p.ln_mean, p.ln_rstd, PROJ && p.tap_y != nullptr);
        MLP_STAMP(7)
    }
#undef MLP_RS_Y
#undef MLP_RS_X
#undef MLP_RS_O
#undef MLP_RS_N
    glds_wait_all();                       // requested pieces that no tile consumed must not outlive the workgroup's LDS
#ifdef CCD_MLP_LAB
    if (t == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(p.ln_mean)[blockIdx.x * 8 + i] = ph[i];
#endif
}

}  // namespace ccd
