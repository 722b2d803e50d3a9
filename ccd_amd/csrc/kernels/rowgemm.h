// rowgemm.h - products whose output row (E = 128 / 256 / 384 / 512 columns) fits ONE wave's accumulators, with a ROW-WISE
// epilogue: RG_RESID_LN = bias + DropPath + fp32 residual + the next LayerNorm (Block.forward, vision_transformer.py:107-113:
// attn.proj; replaces gemm_row384.h's EPI_RESID_LN), and RG_LNBWD = the data gradient of qkv / fc1 with the
// LayerNorm backward pass of those rows as the epilogue (autograd of nn.LayerNorm in Block.forward,
// vision_transformer.py:99,103,107-113):
//     dy = A . W^T  (never written);  xhat = (x - mean) * rstd
//     dx = rstd * (dy*gamma - mean_row(dy*gamma) - xhat * mean_row(dy*gamma*xhat));   g = (accumulate ? g : 0) + dx
//     dgamma += colsum(dy * xhat);  dbeta += colsum(dy);  optional: gb (bf16) = g * rowscale[sample], dbias += colsum(gb)
// Same "row owner" structure as mlp_fused.h (whose second product this is, with the activations coming from HBM instead of
// from the GELU): one workgroup = 4 waves = 128 rows, a wave owns 32 rows and all E columns of them (E/2 accumulator
// registers), the weights stream through the 5-slot LDS ring in pieces of [E/2 output columns][64 k] that all four waves
// consume in lock step, and every lane reads ITS row of A straight into MFMA operand registers (128 contiguous bytes per
// row and 64-wide k-block, R - 1 blocks ahead of their use, across tile boundaries).  Compared with gemm_row384.h
// (one 128 x 384 tile per 8 waves, both operands through LDS) the LDS carries the weights only, the row statistics need
// one shuffle, and the epilogue runs on accumulators that never left their registers.
//
// VMEM bookkeeping (vmcnt counts every VMEM instruction of the wave, in issue order for loads): every piece "window"
// issues exactly KT ring requests (MFMA steps 1, 5, 9, ...) and 2 activation loads (steps 0 and 2), so "my quarter of piece q
// has landed" is vmcnt <= 3 * (KT + 2): at least that many instructions are younger than the last request of piece q, and
// whatever the epilogue adds in between only makes the wait conservative.
// The ACTIVATION loads are hand-tracked too (round 3, prelude: buf_load16_late / vm_arrived): left to the compiler, each group
// of R blocks carried two `s_waitcnt vmcnt(0)` - the ring and every block in flight drained twice per six windows, 4.9 us per
// group where the MFMAs need 1.9.  Quarter j of a block is loaded R - 1 blocks ahead (window hh = j / 2, step 0 or 2) and
// first used at step j * NTH of its block's first window; rg_younger() counts what is issued in between.
#pragma once

namespace ccd {

struct RowGemmParams {
    const bf16_t* A;        // [M, K] bf16 activations (the upstream gradient)
    long lda;
    const bf16_t* W;        // [E, K] bf16: out = A . W^T
    long ldw;
    int M, K;
    const float* x;         // [M, E] fp32: the LayerNorm's input
    long ldx;
    const float* mean;      // [M]
    const float* rstd;
    const float* gamma;     // [E]
    void* g;                // [M, E] gradient of the residual stream (in / out): fp32, or bf16 in the G16 instantiations (round 6:
    long ldg;               // half the bytes of the stream that every LayerNorm-backward epilogue reads and rewrites)
    int accumulate;
    float* dgamma;          // [E] +=
    float* dbeta;
    bf16_t* gb;             // optional [M, E] bf16 = g_new * rowscale
    long ld_gb;
    const float* rowscale;
    int rows_per_sample;
    float* dbias;           // optional [E] += colsum(gb)
    // EPI RG_RESID_LN:  out (f32) = resid + (A . W^T + bias) * rowscale[sample];  ln_y (bf16) = LayerNorm(out), mean / rstd saved
    const float* bias;      // [E] or null
    const float* resid;     // [M, E] fp32
    long ldr;
    float* out;             // [M, E] fp32
    long ldc;
    const float* ln_beta;   // (ln gamma = `gamma`)
    float ln_eps;
    bf16_t* ln_y;
    long ld_y;
    float* ln_mean;
    float* ln_rstd;
    int lab;                // experiment switch (policy key "lab"): n > 0 delays odd workgroups by ~n * 8 k cycles
    // TAP (round 6): a SECOND LayerNorm of the same rows - a segmentation tap (vision_transformer.py:245-249, norm_seg: other gamma /
    // beta, same statistics) - whose backward pass ran as a launch of its own (ln_bwd: d_tap, x, g read, g and gb written: 0.19 ms,
    // three taps per step).  LayerNorm's backward is linear in dy * gamma: dx = core(dy1 * gamma1 + d_tap * gamma_tap), so the tap's
    // gradient joins the product's accumulators in pass A (one more bf16 row stream) and x, g, gb move once.
    const bf16_t* tap_dy;   // [M, E] bf16: gradient of the tap's output, or unused (TAP instantiations only)
    long ld_tap;
    const float* tap_gamma; // [E]
    float* tap_dgamma;      // [E] +=
    float* tap_dbeta;
};

constexpr int RG_SCRATCH = 4096, RG_THREADS = 256, RG_BM = 128;
constexpr int RG_LNBWD = 0, RG_RESID_LN = 1;
// ---- ADMA (round 4): the activation rows arrive by LDS-DMA too.
// Measured on the 8-wave experiment of this round (profiles/r04_rowgemm8_lab.jsonl): a row-per-lane `buffer_load_dwordx4` (every lane
// 16 bytes of ITS row: 64 separate requests) costs ~61 cycles of the CU's vector-memory address path per instruction, a DMA of
// 8 rows x 128 contiguous bytes ~16 - and this kernel issued 8 of the former per 24-MFMA window (a third of the window's time).
// With ADMA a wave moves the next k-block of ITS 32 rows as 4 such DMA instructions (buffer form: rows behind M arrive as zeros) into a
// private 3-slot image ([32 rows][128 B], the pieces' swizzle) and reads the four B-operand fragments of a block from there - one
// window ahead of the block, between the MFMAs, on the same in-order LDS queue as the weight fragments (mlp_product's Extra count).
// Nobody else reads those rows: the wave's own counted vmcnt orders the reads behind the DMA, no barrier.  LDS: the weight ring
// shrinks to 4 slots (96 KiB) for the 48 KiB of activation images; the epilogue's scratch image is the ring slot of the piece
// consumed last (free until the next tile's first window refills it; one barrier per tile in front of its first use).
// VMEM per window: hh = 0 issues the 4 activation DMAs (steps 0, 2, 4, 6: block b + 3, slot b % 3) and KT ring requests (steps
// 1, 5, ...), hh = 1 the KT requests only: any two consecutive windows issue 2 KT + 4, the ring's counted wait.
__host__ __device__ inline int rg_smem_bytes_adma(int E, bool tap = false) { return 4 * mlp_piece_bytes(E) + 4 * 3 * 4096 + (tap ? 7 : 4) * E * 4; }
struct RgAdmaExtra {       // window hh = 1: the next block's four fragment reads ride behind MFMA steps 2, 6, 10, 14
    static constexpr int at(int k) { return (k % 4 == 2 && k < 16) ? 1 : 0; }
};
// ring slots: 5 (four pieces ahead) where 160 KiB allow it, 4 at E = 512 (32-KiB pieces)
__host__ __device__ constexpr int rg_slots(int E) { return E <= 384 ? 5 : 4; }
__host__ __device__ inline int rg_smem_bytes(int E, bool tap = false) {
    return rg_slots(E) * mlp_piece_bytes(E) + 4 * RG_SCRATCH + (tap ? 7 : 4) * E * 4;   // ring, scratch, gamma + 3 vectors (sums | beta, bias) [+ the tap's gamma and 2 sums]
}

// activation k-blocks a lane holds (the newest arrives R - 1 blocks = 2 (R - 1) pieces ahead of its use): 48 registers at
// E = 384 - with 96 the epilogue (accumulators + the next tile's blocks + its own streams) spills.  K / 64 % R == 0.
__host__ __device__ constexpr int rg_ring(int E) { return E == 384 ? 3 : 2; }
// loads + ring requests issued after the load of quarter j (of the block R - 1 ahead) and before its first use.  A window issues,
// in this order: load A (step 0), request 0 (step 1), load B (step 2), requests 1 .. KT - 1 (steps 5, 9, ...); an MFMA step's
// own filler comes after its MFMA.
__host__ __device__ constexpr int rg_younger(int j, int KT, int R) {
    const int win = KT + 2, s = j * KT;                          // first use: step j * NTH (NTH == KT) of window (block, 0)
    const int rest = (j & 1) ? KT - 1 : win - 1;                 // what the load's own window issues after it
    const int between = 2 * (R - 1) - j / 2 - 1;                 // whole windows in between
    const int req = (s + 2) / 4 < KT ? (s + 2) / 4 : KT;         // requests m with 4 m + 1 < s
    return rest + between * win + (s > 0) + (s > 2) + (s > 0 ? req : 0);
}
// (dy * gamma, xhat) in one register between the two epilogue passes
__device__ __forceinline__ unsigned rg_pack(float dg, float xh) {
    const unsigned hi = (__builtin_bit_cast(unsigned, dg) + 0x800u) & 0xFFFFF000u;
    float q = fmaf(xh, 102.4f, 2048.5f);
    q = q < 0.f ? 0.f : (q > 4095.f ? 4095.f : q);
    return hi | (unsigned)q;
}
__device__ __forceinline__ float rg_unpack_dg(unsigned pk) { return __builtin_bit_cast(float, pk & 0xFFFFF000u); }
__device__ __forceinline__ float rg_unpack_xh(unsigned pk) { return ((float)(int)(pk & 0xFFFu) - 2048.f) * (1.0f / 102.4f); }
// one halving step of a sum over lanes: every lane keeps one of (a, b) - the one its bit MASK of the lane id selects - and
// adds the partner lane's copy of it.  16 values -> 8 -> 4 -> 2 -> 1 sums 16 values over 16 lanes in 15 such steps.
template <int MASK>
__device__ __forceinline__ float rg_fold(float a, float b, bool up) {
    const float keep = up ? b : a, send = up ? a : b;
    return keep + lane_xor<MASK>(send);
}
// 16 values per lane -> the sum over 16 lanes (those that differ in bits 0-3 of lq) of value number lq & 15, in every lane
__device__ __forceinline__ float rg_fold16(const float (&v)[16], int lq) {
    float a8[8], a4[4], a2[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) a8[j] = rg_fold<1>(v[2 * j], v[2 * j + 1], (lq & 1) != 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) a4[j] = rg_fold<2>(a8[2 * j], a8[2 * j + 1], (lq & 2) != 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) a2[j] = rg_fold<4>(a4[2 * j], a4[2 * j + 1], (lq & 4) != 0);
    return rg_fold<8>(a2[0], a2[1], (lq & 8) != 0);
}
// v[4 g + e] = this lane's row, column 8 g + 4 hf + e of a 32-column tile: column sums over the 32 rows of the wave, added
// to dst[32 columns] (LDS; the 16 low lanes of each half wave own one column each)
__device__ __forceinline__ void rg_colsum16(const float (&v)[16], float* dst, int lq, int hf) {
    atomicAdd(dst + 8 * ((lq >> 2) & 3) + 4 * hf + (lq & 3), rg_fold16(v, lq));
}

template <int E, int R, int EPI, bool ADMA = false, bool G16 = false, bool TAP = false>
__global__ __launch_bounds__(RG_THREADS, 1) void rowgemm_kernel(RowGemmParams p) {
    static_assert(!G16 || EPI == RG_LNBWD, "the bf16 stream is the LayerNorm-backward epilogue's");
    static_assert(!TAP || EPI == RG_LNBWD, "the tap is a second LayerNorm backward of the same rows");
    constexpr int RG_NSLOT = ADMA ? 4 : rg_slots(E);
    static_assert(!ADMA || (E == 384 && R == 3), "activation DMA: 3 image slots per wave, 160 KiB of LDS at E = 384");
    constexpr int KT = E / 64;             // ring requests (1 KiB wave instructions) per wave and piece
    constexpr int NT = E / 32, NTH = NT / 2;
    constexpr int PIECE = mlp_piece_bytes(E);
    constexpr int AHEAD = RG_NSLOT - 1;
    constexpr int DEPTH = 6;               // fragment reads in flight ahead of their MFMA
    constexpr int NSTEP = 4 * NTH;         // MFMAs per piece
    constexpr int WIN_VM = KT + 2;         // VMEM instructions per window (ADMA: KT + 4 and KT in turn, the same per pair)
    constexpr bool AB_LATE = E <= 384 && !ADMA;     // hand-tracked activation loads into accumulator registers; at E = 512 the accumulators
                                           // fill that file: the blocks stay compiler-managed loads there (with its drains)
    static_assert(E % 128 == 0 && NSTEP == 4 * KT && AHEAD >= 2, "ring bookkeeping");
    static_assert(rg_younger(0, KT, R) <= 63 && rg_younger(1, KT, R) <= 63 && rg_younger(2, KT, R) >= 0 && rg_younger(3, KT, R) >= 0,
                  "vmcnt is a 6-bit counter");
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, hf = lane >> 5, lq = lane & 31;
    const int w = uniform_i32(t >> 6);
    char* scratch = smem + RG_NSLOT * PIECE + w * RG_SCRATCH;       // (ADMA: re-pointed per tile, see the epilogue)
    char* aring = smem + RG_NSLOT * PIECE + w * (3 * 4096);         // ADMA: this wave's activation images
    float* vga = reinterpret_cast<float*>(smem + RG_NSLOT * PIECE + (ADMA ? 4 * 3 * 4096 : 4 * RG_SCRATCH));
    float* cs = vga + E;                   // RG_LNBWD: [3][E] dgamma, dbeta, dbias of this workgroup
    float* vbe = vga + E;                  // RG_RESID_LN: beta, bias
    float* vbi = vbe + E;
    float* vgt = cs + 3 * E;               // TAP: the tap's gamma, then [2][E] its dgamma, dbeta of this workgroup
    float* cst = vgt + E;
    for (int i = t; i < E; i += RG_THREADS) {
        vga[i] = p.gamma[i];
        if (EPI == RG_LNBWD) { cs[i] = 0.f; cs[E + i] = 0.f; cs[2 * E + i] = 0.f; }
        else { vbe[i] = p.ln_beta[i]; vbi[i] = p.bias ? p.bias[i] : 0.f; }
        if constexpr (TAP) { vgt[i] = p.tap_gamma[i]; cst[i] = 0.f; cst[E + i] = 0.f; }
    }
    __syncthreads();

    const int NB = p.K / 64, NP = 2 * NB;  // k-blocks, pieces per row tile (NB % R == 0)
#ifdef CCD_MLP_LAB      // lab build only: cycle totals of wave 0 per phase -> first 64 bytes per workgroup of g (destroyed)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define RG_STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tprev; tprev = tn_; }
#define RG_LAB_KEEP(bit) (!(p.lab & (bit)))   // lab switches of the main loop: 16 = no activation loads, 32 = no ring requests
#else
#define RG_STAMP(i)
#define RG_LAB_KEEP(bit) true
#endif
    const int tiles = (p.M + RG_BM - 1) / RG_BM, G = gridDim.x;

    // ---- weight ring (mlp_fused.h: W2-type pieces): piece (blk, half) = rows half * E/2 + 32 i + 8 w + (0..7), k = 64 blk ..
    const int dr = lane >> 3, dp = lane & 7, drow = 8 * w + dr;
    const unsigned req_lane = (unsigned)(2 * drow) * (unsigned)p.ldw + (unsigned)((dp ^ mlp_swz(drow)) * 16);
    const long req_step_a = 64 * p.ldw, req_step_b = 128 * p.ldw;
    int slot_i = 0, slot_c = 0, pos_i = 0;
    const char* req_base = nullptr;
    char* req_lds = nullptr;
    auto issue_prepare = [&]() {
        const int blk = pos_i >> 1, half = pos_i & 1;
        req_base = reinterpret_cast<const char*>(p.W) + ((long)(half * (E / 2)) * p.ldw + 64 * blk) * 2;
        req_lds = smem + slot_i * PIECE + w * 1024;
        slot_i = slot_i + 1 == RG_NSLOT ? 0 : slot_i + 1;
        pos_i = pos_i + 1 == NP ? 0 : pos_i + 1;
    };
    auto issue_one = [&](int i) {
        glds16(req_base + ((i & 1) * req_step_a + (i >> 1) * req_step_b) + req_lane, req_lds + 4096 * i);
    };
    const unsigned smem_addr = lds_addr_of(smem);
    auto acquire = [&]() -> unsigned {
        glds_wait<ADMA ? (AHEAD - 1) * (KT + 2) : (AHEAD - 1) * WIN_VM>();     // (ADMA, AHEAD - 1 = 2: one window of each kind)
        RG_STAMP(1)
        lds_barrier();
        RG_STAMP(2)
        issue_prepare();
        const unsigned sb = smem_addr + (unsigned)(slot_c * PIECE);
        slot_c = slot_c + 1 == RG_NSLOT ? 0 : slot_c + 1;
        return sb;
    };
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
        issue_prepare();
#pragma unroll
        for (int i = 0; i < KT; ++i) issue_one(i);
    }
    unsigned off2[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) off2[kk] = (unsigned)(lq * 128 + (((2 * kk + hf) ^ mlp_swz(lq)) * 16));

    const buf_desc ds_a = make_desc(p.A, (unsigned)((((long)p.M - 1) * p.lda + p.K) * 2));
    const buf_rsrc rs_a = make_rsrc(p.A, (unsigned)((((long)p.M - 1) * p.lda + p.K) * 2));
    // the row-wise streams of the epilogue: (x, g, gb) for the LayerNorm backward, (resid, out, ln_y) for residual + LayerNorm
    const buf_rsrc rs_x = EPI == RG_LNBWD ? make_rsrc(p.x, (unsigned)((((long)p.M - 1) * p.ldx + E) * 4))
                                          : make_rsrc(p.resid, (unsigned)((((long)p.M - 1) * p.ldr + E) * 4));
    const buf_rsrc rs_g = EPI == RG_LNBWD ? make_rsrc(p.g, (unsigned)((((long)p.M - 1) * p.ldg + E) * (G16 ? 2 : 4)))
                                          : make_rsrc(p.out, (unsigned)((((long)p.M - 1) * p.ldc + E) * 4));
    const buf_rsrc rs_b = EPI == RG_LNBWD ? make_rsrc(p.gb, p.gb ? (unsigned)((((long)p.M - 1) * p.ld_gb + E) * 2) : 0u)
                                          : make_rsrc(p.ln_y, (unsigned)((((long)p.M - 1) * p.ld_y + E) * 2));
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
    // this lane's row of A: k-block b of row tile tt, quarter j = the B operand of k-step j (k = 64 b + 16 j + 8 hf .. + 7)
    const unsigned lo_a = (unsigned)((lq * p.lda + 8 * hf) * 2);
    u32x4 ab[R][4];
    auto load_a = [&](u32x4& dst, int tt, int b, int j) {
        const unsigned so = (unsigned)(tt * RG_BM + 32 * w) * (unsigned)(p.lda * 2) + (unsigned)(128 * b + 32 * j);
        if constexpr (AB_LATE) buf_load16_late(dst, ds_a, lo_a, so);
        else dst = buf_load16(rs_a, lo_a, so);
    };
    // ADMA: instruction jj of a block = rows 8 jj .. + 7 of the wave, 128 bytes each; LDS position dp of image row r holds the
    // logical slot dp ^ swz(r) (the fragment reads are the pieces': off2)
    unsigned lane_dma[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
        lane_dma[jj] = (unsigned)(((8 * jj + (lane >> 3)) * p.lda) * 2 + (((lane & 7) ^ mlp_swz(8 * jj + (lane >> 3))) * 16));
    auto dma_a = [&](int jj, int tt, int b, int slot) {
        stream_bufdma16<NT_RG_A>(rs_a, lane_dma[jj], (unsigned)(tt * RG_BM + 32 * w) * (unsigned)(p.lda * 2) + (unsigned)(128 * b), aring + slot * 4096 + jj * 1024);
    };
    bf16x8 fr[3][4];                       // ADMA: B-operand fragments of the block with index % 3 == slot (k-step j = fr[.][j])
    const unsigned aring_addr = lds_addr_of(aring);
    if constexpr (ADMA) {
        if ((int)blockIdx.x < tiles) {
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) dma_a(jj, blockIdx.x, b, b);
        }
        glds_wait_all();
        wave_lds_fence();                  // (no instruction: the wave's vmcnt covers all of its lanes; the CPU executor runs lanes as fibers)
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_read_frag<0>(fr[0][j], aring_addr + off2[j]);
        lds_drain();
        lds_landed4(fr[0]);
    } else {
    if ((int)blockIdx.x < tiles) {
#pragma unroll
        for (int b = 0; b < R - 1; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) load_a(ab[b][j], blockIdx.x, b, j);
    }
    // the counted waits of the main loop assume a full pipeline behind them: start from a drained queue (once per workgroup)
    glds_wait_all();
    }
    const float inv_e = 1.0f / (float)E;
    if (p.lab > 0 && p.lab < 16 && (blockIdx.x & 1)) wave_sleep(p.lab);

    for (int tile = blockIdx.x; tile < tiles; tile += G) {
        const int m0 = tile * RG_BM, r0 = m0 + 32 * w;
        const int row = r0 + lq, grow = row < p.M ? row : p.M - 1;
        float mu = 0.f, rs = 0.f;
        if (EPI == RG_LNBWD) { mu = p.mean[grow]; rs = p.rstd[grow]; }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        RG_STAMP(0)
#pragma unroll 1
        for (int grp = 0; grp < NB / R; ++grp) {
            mlp_static_for<0, R>([&](auto I) {
                constexpr int i = decltype(I)::value, sp = (i + R - 1) % R;
                // the block that slot sp receives while block (grp, i) is multiplied: R - 1 blocks ahead, maybe of the next tile
                int tb = grp * R + i + R - 1, tt = tile;
                if (tb >= NB) { tb -= NB; tt += G; }
                if (tt >= tiles) tt = tile;        // behind the last tile: re-read this one (an out-of-range load of the whole wave
                                                   // need not keep its place in the return order)
                if constexpr (ADMA) {
                    // block (grp, i): fragments in fr[i]; block + 3 (maybe of the next tile) goes into image slot i in window 0,
                    // block + 1's fragments come out of image slot (i + 1) % 3 in window 1
                    int tb3 = grp * R + i + 3, tt3 = tile;
                    if (tb3 >= NB) { tb3 -= NB; tt3 += G; }
                    if (tt3 >= tiles) tt3 = tile;
                    {
                        const unsigned sb = acquire();
                        const unsigned areg[4] = {sb + off2[0], sb + off2[1], sb + off2[2], sb + off2[3]};
                        mlp_product<NSTEP, DEPTH, MlpMapP2<NTH>, MlpNoExtra>(
                            areg,
                            [&](auto K, const bf16x8& a) {
                                constexpr int k = decltype(K)::value;
                                acc[k % NTH] = mfma_32x32x16_bf16(a, fr[i][k / NTH], acc[k % NTH]);
                            },
                            [&](auto K) {
                                constexpr int k = decltype(K)::value;
                                if constexpr (k % 4 == 1) issue_one(k / 4);
                                if constexpr (k % 2 == 0 && k < 8) dma_a(k / 2, tt3, tb3, i);
                            });
                        RG_STAMP(3)
                    }
                    {
                        const unsigned sb = acquire();
                        const unsigned areg[4] = {sb + off2[0], sb + off2[1], sb + off2[2], sb + off2[3]};
                        constexpr int nx = (i + 1) % 3;
                        mlp_product<NSTEP, DEPTH, MlpMapP2<NTH>, RgAdmaExtra>(
                            areg,
                            [&](auto K, const bf16x8& a) {
                                constexpr int k = decltype(K)::value;
                                acc[NTH + k % NTH] = mfma_32x32x16_bf16(a, fr[i][k / NTH], acc[NTH + k % NTH]);
                            },
                            [&](auto K) {
                                constexpr int k = decltype(K)::value;
                                if constexpr (k % 4 == 1) issue_one(k / 4);
                                if constexpr (RgAdmaExtra::at(k) != 0) lds_read_frag<0>(fr[nx][k / 4], aring_addr + (unsigned)(nx * 4096) + off2[k / 4]);
                            });
                        lds_landed4(fr[nx]);       // (the window's last fragment wait drained the queue: every extra read sits in front of read 23)
                        RG_STAMP(3)
                    }
                } else
                mlp_static_for<0, 2>([&](auto HH) {
                    constexpr int hh = decltype(HH)::value;
                    const unsigned sb = acquire();
                    const unsigned areg[4] = {sb + off2[0], sb + off2[1], sb + off2[2], sb + off2[3]};
                    mlp_product<NSTEP, DEPTH, MlpMapP2<NTH>, MlpNoExtra>(
                        areg,
                        [&](auto K, const bf16x8& a) {
                            constexpr int k = decltype(K)::value;
                            if constexpr (AB_LATE && hh == 0 && k % NTH == 0) vm_arrived<rg_younger(k / NTH, KT, R)>(ab[i][k / NTH]);
                            acc[NTH * hh + k % NTH] = mfma_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, ab[i][k / NTH]), acc[NTH * hh + k % NTH]);
                        },
                        [&](auto K) {
                            constexpr int k = decltype(K)::value;
                            if constexpr (k % 4 == 1) { if (RG_LAB_KEEP(32)) issue_one(k / 4); }
                            if constexpr (k == 0) { if (RG_LAB_KEEP(16)) load_a(ab[sp][2 * hh], tt, tb, 2 * hh); }
                            if constexpr (k == 2) { if (RG_LAB_KEEP(16)) load_a(ab[sp][2 * hh + 1], tt, tb, 2 * hh + 1); }
                        });
                    RG_STAMP(3)
                });
            });
        }
        // ---- epilogue: a row is complete inside its two lanes (lane, lane ^ 32); acc[nt][4 g + e] = column 32 nt + 8 g + 4 hf + e.
        if constexpr (ADMA) {
            // the scratch image: the ring slot of the piece consumed last - every wave is done with it behind this barrier, and its
            // refill is issued behind the next tile's first barrier
            lds_barrier();
            scratch = smem + (slot_c == 0 ? RG_NSLOT - 1 : slot_c - 1) * PIECE + w * RG_SCRATCH;
        }
        if constexpr (EPI == RG_RESID_LN) {
            // (mlp_fused.h's epilogue)  Pass A: out = x + (acc + bias) * sc back into the accumulators, LayerNorm statistics
            float sc = 1.0f;
            if (p.rowscale) sc = p.rowscale[grow / p.rows_per_sample];
            float s1 = 0.f, s2 = 0.f;
            // (the vector tables are addressed from a per-tile opaque lane id: from the loop-invariant one the optimiser hoists
            // one address register per (tile, group) out of the persistent loop - 80 spilled registers, and every reload of a
            // spilled register is a scratch load behind `s_waitcnt vmcnt(0)`: a drain of the stores this epilogue streams)
            const LaneOff lo(t);
            const float* vbi_h = vbi + 4 * lo.hf;
            const float* vga_h = vga + 4 * lo.hf;
            const float* vbe_h = vbe + 4 * lo.hf;
            {
                const unsigned so = (unsigned)r0 * (unsigned)(p.ldr * 4), lo_x = lo.frag(p.ldr, 4, 4);
                constexpr int PA = 4;
                u32x4 xb[PA][4];
                auto load_x = [&](int nt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) xb[nt % PA][g] = buf_load16(rs_x, lo_x, so + (32 * nt + 8 * g) * 4);
                };
#pragma unroll
                for (int nt = 0; nt < PA - 1 && nt < NT; ++nt) load_x(nt);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (nt + PA - 1 < NT) load_x(nt + PA - 1);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4v x = __builtin_bit_cast(f32x4v, xb[nt % PA][g]);
                        const f32x4v b = *reinterpret_cast<const f32x4v*>(vbi_h + 32 * nt + 8 * g);
                        const float xx[4] = {x.x, x.y, x.z, x.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = xx[e] + (acc[nt][4 * g + e] + bb[e]) * sc;
                            acc[nt][4 * g + e] = v;
                            s1 += v;
                            s2 = fmaf(v, v, s2);
                        }
                    }
                    CCD_SCHED_FENCE();
                    asm volatile("" ::: "memory");
                }
            }
            s1 += shfl_xor(s1, 32);
            s2 += shfl_xor(s2, 32);
            const float mean = s1 * inv_e;
            float var = s2 * inv_e - mean * mean;
            var = var > 0.f ? var : 0.f;
            const float rstd = 1.0f / sqrtf(var + p.ln_eps);
            if (hf == 0 && row < p.M) { p.ln_mean[row] = mean; p.ln_rstd[row] = rstd; }
            const unsigned lo_o = lo.rows8(p.ldc, 4), lo_n = lo.rows8(p.ld_y, 2);
            // Pass B: out (fp32, one 32-column tile at a time) and ln_y (bf16, two tiles) through the scratch image
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                u32x2 ypk[2][4];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int nt = 2 * np + tt;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float v[4] = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                        *reinterpret_cast<f32x4v*>(scratch + lo.scr_wr(2 * g + lo.hf)) = f32x4v{v[0], v[1], v[2], v[3]};
                        const int n = 32 * nt + 8 * g;
                        const f32x4v ga = *reinterpret_cast<const f32x4v*>(vga_h + n), be = *reinterpret_cast<const f32x4v*>(vbe_h + n);
                        ypk[tt][g].x = pack_bf2((v[0] - mean) * rstd * ga.x + be.x, (v[1] - mean) * rstd * ga.y + be.y);
                        ypk[tt][g].y = pack_bf2((v[2] - mean) * rstd * ga.z + be.z, (v[3] - mean) * rstd * ga.w + be.w);
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                        buf_store16(rs_g, lo_o, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldc * 4) + 128 * nt, o);
                    }
                    wave_lds_fence();
                    CCD_SCHED_FENCE();
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<u32x2*>(scratch + lo.scr_wr(4 * tt + g) + 8 * lo.hf) = ypk[tt][g];
                wave_lds_fence();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                    buf_store16(rs_b, lo_n, (unsigned)(r0 + 8 * i) * (unsigned)(p.ld_y * 2) + 128 * np, o);
                }
                wave_lds_fence();
            }
        } else {
        // Pass A: the two row means and the column sums of dy * xhat and dy (x streams in two tiles ahead of its use).
        const unsigned so_x = (unsigned)r0 * (unsigned)(p.ldx * 4);
        float s1 = 0.f, sq = 0.f;
        {
            const unsigned lo_x = LaneOff(t).frag(p.ldx, 4, 4);
            constexpr int PA = 4;          // tiles of x in flight (HBM latency x 14 GB/s per CU and tile in flight)
            u32x4 xb[PA][4];
            // TAP: the lane's four bf16 of d_tap per column group, the same tiles ahead
            const buf_rsrc rs_t = make_rsrc(TAP ? p.tap_dy : nullptr, TAP ? (unsigned)((((long)p.M - 1) * p.ld_tap + E) * 2) : 0u);
            const unsigned lo_t = TAP ? LaneOff(t).frag(p.ld_tap, 2, 4) : 0u, so_t = TAP ? (unsigned)r0 * (unsigned)(p.ld_tap * 2) : 0u;
            buf_u32x2 tb[TAP ? PA : 1][4];
            auto load_x = [&](int nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    xb[nt % PA][g] = stream_load16<NT_RG_X>(rs_x, lo_x, so_x + (32 * nt + 8 * g) * 4);
                    if constexpr (TAP) tb[nt % PA][g] = buf_load8(rs_t, lo_t, so_t + (32 * nt + 8 * g) * 2);
                }
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
                        const float dy = acc[nt][4 * g + e], xh = (xx[e] - mu) * rs;
                        float dg = dy * gg[e];
                        if constexpr (TAP) {       // the tap's share of dy * gamma (its own column sums follow below, on tb again)
                            const unsigned wd = (e < 2) ? tb[nt % PA][g].x : tb[nt % PA][g].y;
                            const float dt = (e & 1) ? bf_hi(wd) : bf_lo(wd);
                            dg = fmaf(dt, vgt[32 * nt + 8 * g + 4 * hf + e], dg);
                        }
                        // what pass B needs, in the accumulator's own register, so that x is read once: dy * gamma rounded to 12
                        // mantissa bits (the unfused pair rounds dy to 8) above xhat in 12-bit fixed point (|xhat| < 20, steps of 0.01;
                        // it only scales the small second-moment correction)
                        // (opaque: the optimiser sinks the packing into pass B otherwise, and carries BOTH values there)
                        acc[nt][4 * g + e] = opaque_f32(__builtin_bit_cast(float, rg_pack(dg, xh)));
                        s1 += dg;
                        sq = fmaf(dg, xh, sq);
                        vg[4 * g + e] = dy * xh;
                        vb[4 * g + e] = dy;
                    }
                }
                rg_colsum16(vg, cs + 32 * nt, lq, hf);
                rg_colsum16(vb, cs + E + 32 * nt, lq, hf);
                if constexpr (TAP) {           // dgamma_tap += colsum(d_tap * xhat), dbeta_tap += colsum(d_tap): xhat back out of the packed word
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned wd = (e < 2) ? tb[nt % PA][g].x : tb[nt % PA][g].y;
                            const float dt = (e & 1) ? bf_hi(wd) : bf_lo(wd);
                            const f32x4v xv = __builtin_bit_cast(f32x4v, xb[nt % PA][g]);
                            const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
                            vg[4 * g + e] = dt * ((xx[e] - mu) * rs);
                            vb[4 * g + e] = dt;
                        }
                    rg_colsum16(vg, cst + 32 * nt, lq, hf);
                    rg_colsum16(vb, cst + E + 32 * nt, lq, hf);
                }
                CCD_SCHED_FENCE();             // one tile at a time: interleaved tiles (and every load of the pass hoisted to
                asm volatile("" ::: "memory");   // its top) cost more registers than there are
            }
        }
        RG_STAMP(4)
        s1 += shfl_xor(s1, 32);
        sq += shfl_xor(sq, 32);
        const float c1 = rs * s1 * inv_e, c2 = rs * sq * inv_e;
        float sc = 1.0f;
        if (p.gb && p.rowscale) sc = p.rowscale[grow / p.rows_per_sample];
        // Pass B: dx, g and gb per 32-column tile, leaving through the wave's scratch image as 128-byte row segments
        {
            const LaneOff lo(t);
            constexpr int GB = G16 ? 2 : 4;            // bytes per element of the gradient stream
            const unsigned lo_gl = lo.frag(p.ldg, GB, 4), so_g = (unsigned)r0 * (unsigned)(p.ldg * GB);
            const unsigned lo_o = lo.rows8(p.ldg, GB), lo_n = lo.rows8(p.ld_gb, 2);
            constexpr int PB = 3;
            u32x4 gbuf[PB][4];                         // (G16: .x / .y hold the lane's four bf16)
            auto load_xg = [&](int nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    gbuf[nt % PB][g] = u32x4{0u, 0u, 0u, 0u};
                    if (p.accumulate) {
                        if constexpr (G16) {
                            const buf_u32x2 h = buf_load8(rs_g, lo_gl, so_g + (32 * nt + 8 * g) * 2);
                            gbuf[nt % PB][g].x = h.x; gbuf[nt % PB][g].y = h.y;
                        } else gbuf[nt % PB][g] = stream_load16<NT_RG_G>(rs_g, lo_gl, so_g + (32 * nt + 8 * g) * 4);
                    }
                }
            };
#pragma unroll
            for (int nt = 0; nt < PB - 1 && nt < NT; ++nt) load_xg(nt);
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                u32x2 ypk[2][4], gpk[2][4];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int nt = 2 * np + tt;
                    if (nt + PB - 1 < NT) load_xg(nt + PB - 1);
                    float vbi[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float oo[4];
                        if constexpr (G16) {
                            oo[0] = bf_lo(gbuf[nt % PB][g].x); oo[1] = bf_hi(gbuf[nt % PB][g].x);
                            oo[2] = bf_lo(gbuf[nt % PB][g].y); oo[3] = bf_hi(gbuf[nt % PB][g].y);
                        } else {
                            const f32x4v go = __builtin_bit_cast(f32x4v, gbuf[nt % PB][g]);
                            oo[0] = go.x; oo[1] = go.y; oo[2] = go.z; oo[3] = go.w;
                        }
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float pf = acc[nt][4 * g + e];      // (bit_cast applied to the element lvalue itself reads the wrong lane)
                            const unsigned pk = __builtin_bit_cast(unsigned, pf);
                            const float dx = fmaf(-c2, rg_unpack_xh(pk), fmaf(rg_unpack_dg(pk), rs, -c1));
                            v[e] = oo[e] + dx;
                        }
                        if constexpr (G16) {
                            gpk[tt][g].x = pack_bf2(v[0], v[1]);
                            gpk[tt][g].y = pack_bf2(v[2], v[3]);
                        } else *reinterpret_cast<f32x4v*>(scratch + lo.scr_wr(2 * g + lo.hf)) = f32x4v{v[0], v[1], v[2], v[3]};
                        if (p.gb) {
                            ypk[tt][g].x = pack_bf2(v[0] * sc, v[1] * sc);
                            ypk[tt][g].y = pack_bf2(v[2] * sc, v[3] * sc);
                            // sum what the GEMMs will read: the bf16-rounded values
                            vbi[4 * g] = bf_lo(ypk[tt][g].x); vbi[4 * g + 1] = bf_hi(ypk[tt][g].x);
                            vbi[4 * g + 2] = bf_lo(ypk[tt][g].y); vbi[4 * g + 3] = bf_hi(ypk[tt][g].y);
                        }
                    }
                    if (p.gb && p.dbias) rg_colsum16(vbi, cs + 2 * E + 32 * nt, lq, hf);
                    if constexpr (!G16) {
                        wave_lds_fence();
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                            stream_store16<NT_RG_G>(rs_g, lo_o, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldg * 4) + 128 * nt, o);
                        }
                        wave_lds_fence();
                    }
                    CCD_SCHED_FENCE();
                    asm volatile("" ::: "memory");
                }
                if constexpr (G16) {       // the bf16 stream leaves like gb: two tiles = one 128-byte row segment
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
                }
                if (p.gb) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<u32x2*>(scratch + lo.scr_wr(4 * tt + g) + 8 * lo.hf) = ypk[tt][g];
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                        stream_store16<NT_RG_GB>(rs_b, lo_n, (unsigned)(r0 + 8 * i) * (unsigned)(p.ld_gb * 2) + 128 * np, o);
                    }
                    wave_lds_fence();
                }
            }
        }
        }
        RG_STAMP(5)
    }
    glds_wait_all();                       // requested pieces that no tile consumed must not outlive the workgroup's LDS
    __syncthreads();
    if (EPI == RG_LNBWD) {
        for (int i = t; i < E; i += RG_THREADS) {
            atomicAdd(p.dgamma + i, cs[i]);
            atomicAdd(p.dbeta + i, cs[E + i]);
            if (p.dbias) atomicAdd(p.dbias + i, cs[2 * E + i]);
            if constexpr (TAP) {
                atomicAdd(p.tap_dgamma + i, cst[i]);
                atomicAdd(p.tap_dbeta + i, cst[E + i]);
            }
        }
    }
#ifdef CCD_MLP_LAB
    RG_STAMP(6)
    if (t == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(EPI == RG_LNBWD ? p.g : (void*)p.out)[blockIdx.x * 8 + i] = ph[i];
#endif
}

}  // namespace ccd
