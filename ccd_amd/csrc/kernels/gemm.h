// gemm.h - bf16 MFMA GEMM family for gfx950 (fp32 accumulate), 128x128 output tile, BK = 64, 4 waves (2x2).
//
//   NT :  C[M,N] (+)= A[M,K] . B[N,K]^T        A, B row-major with the contraction index contiguous
//         (every nn.Linear forward, and every dX = dY . W via the pre-transposed bf16 copy of W)
//   TN :  C[P,Q] (+)= sum_m A[m,P]^T . B[m,Q]   both operands have the contraction index as the ROW index
//         (every dW = dY^T . X), split over blockIdx.z along m with fp32 atomics
//
// LDS tile image (both operands, both modes): [128 rows][64 k] bf16 = 128 B per row, 16-byte slots
// XOR-swizzled by ((row >> 1) & 7): conflict-free for the ds_read_b128 fragment reads (16 distinct rows
// per lane group) and for the staging writes of both loaders (see the loaders).  Two stages (64 KiB), the
// fp32 epilogue staging tile [128][132] aliases them, so 2 workgroups fit a CU's 160 KiB.
// MFMA: v_mfma_f32_32x32x16_bf16, each wave owns a 64x64 quadrant = 2x2 accumulators of 16 VGPRs.
#pragma once

namespace ccd {

enum GemmEpilogue {
    EPI_BF16 = 0,       // C(bf16) = acc + bias
    EPI_GELU = 1,       // C(bf16) = u = acc + bias ; C2(bf16) = gelu(u)
    EPI_RESID = 2,      // C(f32)  = resid + (acc + bias) * rowscale[row / rows_per_sample]
    EPI_F32 = 3,        // C(f32)  = acc + bias
    EPI_ATOMIC = 4,     // C(f32) += acc                       (split-K partial sums)
    EPI_DGELU = 5,      // C(bf16) = acc * gelu'(aux) ; optional C2(bf16) = gelu(aux)   (aux = saved pre-activation u, bf16)
    EPI_BF16_ADDF32 = 6, // C(bf16) = acc + bias ; C2(f32) += acc (unused hook kept for head experiments)
    EPI_RESID_LN = 7,    // EPI_RESID + LayerNorm of the finished rows (gemm_row384.h only)
    EPI_LNBWD = 8        // the product IS dy of a LayerNorm: its backward (g (+)= dx, dgamma, dbeta, optional bf16 tail) in the epilogue
};

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    long lda, ldb;          // row strides in elements
    int M, N, K;            // NT: C is MxN, contraction K (multiple of 64).  TN: C is PxQ = MxN, contraction length K
    void* C;
    long ldc;
    void* C2;
    long ldc2;
    const float* bias;      // [N] or null
    const float* resid;     // EPI_RESID
    long ldr;
    const float* rowscale;  // EPI_RESID: per-sample scale (DropPath), or null
    int rows_per_sample;
    int rps_shift;          // log2(rows_per_sample) when it is a power of two (256 tokens per view), else -1: avoids a
                            // ~40-instruction integer division per output row in the residual epilogue
    const bf16_t* aux;      // EPI_DGELU
    long ldaux;
    int k_per_split;        // TN: contraction rows handled by one blockIdx.z slice (multiple of 64)
    int m_fastest;          // tile order: 0 = column tiles fastest, 1 = row tiles fastest
    int work_items;         // tiles x splits (the persistent grid may be smaller)
    float alpha;            // scales acc before the epilogue
    float* colsum;          // optional [N] fp32: += column sums of the (final) output tile, e.g. the bias gradient
    float* colsum_a;        // TN only, optional [P] fp32: += column sums of operand A (the bias gradient that belongs to dW = dY^T X)
    // ---- gemm_tn384.h only: an optional SECOND weight-gradient problem over the same contraction rows, C2[M2,N2] += A2^T . B2
    // (C2 / ldc2 above), sharing the launch - and the contraction slices - with the first
    const bf16_t* A2;
    const bf16_t* B2;
    long lda2, ldb2;
    int M2, N2;
    // the workgroups of XCD x are units1[x] groups of the first problem's tiles, then units2[x] groups of the second's (one byte
    // per XCD); a group works on one contraction slice of ITS problem: slices2 slices of per2 rows for the second problem
    unsigned long long units1, units2;
    int slices2, per2;
    // optional split-K workspaces (round 4): slice s of a problem STORES its tile into ws[(s * P + row) * Q + col] instead of adding it
    // to C with fp32 atomics; tn3_reduce_kernel then adds the slices' sum to C in one pass.  null = atomic epilogue
    float* ws;
    float* ws2;
    float* colsumsq;        // optional [N] fp32: += column sums of squares (BatchNorm batch statistics), EPI_BF16 only
    // ---- implicit-GEMM convolution (NT, GATHER instantiation): A row r is pixel (n, oy, ox) of a 2^gh x 2^gw grid,
    // contraction index k = tap * cin + c reads source pixel (oy*s_mul + dy(tap), ox*s_mul + dx(tap)) of an s_h x s_w
    // image with `lda` channels per pixel (zero outside); covers 3x3 conv, its data gradient, the four parity classes
    // of a 4x4/stride-2 transposed conv and that layer's data gradient.
    int g_h_log2, g_w_log2, s_h, s_w, s_mul, cin;
    unsigned long long dy_pack, dx_pack;   // tap offsets as 4-bit fields (d + 8), tap i in bits 4i..4i+3: register lookup
    int c_map, c_py, c_px;  // c_map: output row (n, oy, ox) -> (n, 2*oy + c_py, 2*ox + c_px) of the 2x upsampled grid
    // ---- LayerNorm of the output rows folded into the residual epilogue (full-row kernel gemm_row384.h only):
    // y = LN(C row) * ln_gamma + ln_beta (bf16, row stride ld_y), statistics for the backward pass
    const float* ln_gamma;
    const float* ln_beta;
    bf16_t* ln_y;
    long ld_y;
    float* ln_mean;
    float* ln_rstd;
    float ln_eps;
    // ---- LayerNorm BACKWARD folded into the epilogue (EPI_LNBWD, gemm_row384.h): acc = dy; x = resid (ldr), statistics
    // ln_mean / ln_rstd (inputs), ln_gamma; C = g (fp32, += dx when lnb_accumulate); optional tail gb = bf16(g_new *
    // rowscale) with lnb_dbias += column sums of gb; lnb_dgamma / lnb_dbeta += column sums
    int lnb_accumulate;
    bf16_t* lnb_gb;
    long ld_gb;
    float* lnb_dgamma;
    float* lnb_dbeta;
    float* lnb_dbias;
    const int* d_rows;      // optional device-side row count: NT rows M / TN contraction length K become
    int rows_mul;           //   min(static value, d_rows[0] * rows_mul); the grid is sized for the static value
};

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64;
constexpr int GEMM_STAGE_BYTES = GEMM_BM * GEMM_BK * 2;          // one operand, one stage: 16 KiB
constexpr int GEMM_CS_LD = 132;                                   // fp32 staging row stride (floats)
constexpr int GEMM_SMEM_BYTES = GEMM_BM * GEMM_CS_LD * 4;         // 67,584 B  (>= 4 stages * 16 KiB)

// 16-byte slot swizzle of the LDS image.  Brute-force checked (all lane groups of ds_read_b128 / ds_write_b128 /
// ds_write_b64): fragment reads and NT staging writes conflict-free, TN staging writes 2-way.
__device__ __forceinline__ int gemm_swz(int row, int slot) { return slot ^ (((row >> 1) ^ (row >> 4)) & 7); }

// ---- operand loaders.  All global reads are 16-byte BUFFER loads (prelude: make_rsrc / buf_load16): the descriptor
// is wave-uniform (base pointer advanced per k-tile in SGPRs), each thread keeps loop-invariant 32-bit byte offsets,
// and every predicate - row beyond the matrix, conv tap outside the image, contraction tail, prefetch past the last
// k-tile - is expressed as an out-of-range offset or an empty descriptor (hardware returns zeros), so the main loop
// carries no exec-mask branches, no 64-bit pointer arithmetic and no zero-fill moves.
//
// NT staging: thread t moves 4 x 16 B per operand; 8 consecutive lanes cover one 128-B tile row (coalesced),
// chunk i of thread t is tile row (t >> 3) + 32 i, slot t & 7.
struct NtLoader {
    unsigned off[4];
    __device__ __forceinline__ void init(long ld, int row0, int nrows) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + (t >> 3) + 32 * i;
            off[i] = row < nrows ? (unsigned)(((long)row * ld + (t & 7) * 8) * 2) : BUF_OOB;
        }
    }
    __device__ __forceinline__ void load(buf_rsrc rs, u32x4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = buf_load16(rs, off[i]);
    }
};
// gather variant of NtLoader for the implicit-GEMM convolutions (see GemmParams): the k-tile (tap, c0) is
// wave-uniform, the tap's source pixel is per row
struct GatherLoader {
    unsigned pix[4];                 // per chunk row: element offset of the image's first pixel + this thread's slot
    int pos[4];                      // (oy*s_mul) << 16 | (ox*s_mul); rows beyond the matrix: a y no image reaches
    __device__ __forceinline__ void init(const GemmParams& p, int row0) {
        const int t = threadIdx.x;
        const int hw = p.g_h_log2 + p.g_w_log2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + (t >> 3) + 32 * i;
            const int n = row >> hw, oy = (row >> p.g_w_log2) & ((1 << p.g_h_log2) - 1), ox = row & ((1 << p.g_w_log2) - 1);
            pix[i] = (unsigned)(n * p.s_h * p.s_w);
            pos[i] = row < p.M ? ((oy * p.s_mul) << 16) | (ox * p.s_mul) : (0x4000 << 16);
        }
    }
    __device__ __forceinline__ void load(const GemmParams& p, buf_rsrc rs, int kt, u32x4 (&r)[4]) const {
        const int k0 = kt * GEMM_BK, tap = (k0 / p.cin) & 15, c0 = k0 % p.cin;      // scalar (kt is wave-uniform)
        const int dy = (int)((p.dy_pack >> (4 * tap)) & 15ull) - 8, dx = (int)((p.dx_pack >> (4 * tap)) & 15ull) - 8;
        const unsigned lane_c = (unsigned)(c0 + (threadIdx.x & 7) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sy = (pos[i] >> 16) + dy, sx = (pos[i] & 0xffff) + dx;
            const bool ok = (unsigned)sy < (unsigned)p.s_h && (unsigned)sx < (unsigned)p.s_w;
            const unsigned o = ((pix[i] + (unsigned)(sy * p.s_w + sx)) * (unsigned)p.lda + lane_c) * 2u;
            r[i] = buf_load16(rs, ok ? o : BUF_OOB);
        }
    }
};
__device__ __forceinline__ void gemm_store_nt(char* tile, const u32x4 (&r)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (t >> 3) + 32 * i, slot = t & 7;
        *reinterpret_cast<u32x4*>(tile + row * 128 + gemm_swz(row, slot) * 16) = r[i];
    }
}
// ---- TN staging: the tile is 64 contraction rows x 128 columns in memory.  Thread t owns column block cb = t & 15
// (8 columns = 16 B) and contraction block mb = t >> 4 (4 rows): 16 consecutive lanes read 256 contiguous bytes of one
// row (coalesced), the 4 x 8 block is transposed in registers, and store j writes tile row 8*cb + j, 8-byte chunk mb.
// The descriptor of k-tile kt starts at that tile's first contraction row and ends at the slice's last one, so the
// contraction tail reads zeros by itself.
struct TnLoader {
    unsigned off[4];
    __device__ __forceinline__ void init(long ld, int col0, int ncols) {
        const int t = threadIdx.x;
        const int col = col0 + 8 * (t & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            off[i] = col < ncols ? (unsigned)(((long)(4 * (t >> 4) + i) * ld + col) * 2) : BUF_OOB;
    }
    __device__ __forceinline__ void load(buf_rsrc rs, u32x4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = buf_load16(rs, off[i]);
    }
};
__device__ __forceinline__ buf_rsrc gemm_tn_rsrc(const bf16_t* base, long ld, int k_begin, int k_end, int kt) {
    const long row = (long)k_begin + (long)kt * GEMM_BK;
    long bytes = ((long)k_end - row) * ld * 2;
    bytes = bytes < 0 ? 0 : (bytes > 0x7ffffff0L ? 0x7ffffff0L : bytes);
    return make_rsrc(base + row * ld, (unsigned)bytes);
}
// gather variant for the B operand: the weight gradient of a convolution contracts over pixels, B row r is the patch
// vector of pixel r (column tap*cin + c), read straight from the source image instead of an im2col buffer.
// A thread's 8 columns lie inside one tap (cin % 8 == 0); its 4 contraction rows are 4 consecutive x positions of one
// image row (grid width % 4 == 0).
struct TnGatherLoader {
    unsigned lane_c;
    int dy, dx;
    bool col_ok;
    __device__ __forceinline__ void init(const GemmParams& p, int col0, int ncols) {
        const int col = col0 + 8 * (threadIdx.x & 15);
        col_ok = col < ncols;
        const int tap = col_ok ? col / p.cin : 0;
        lane_c = (unsigned)(col_ok ? col % p.cin : 0);
        dy = (int)((p.dy_pack >> (4 * tap)) & 15ull) - 8;
        dx = (int)((p.dx_pack >> (4 * tap)) & 15ull) - 8;
    }
    __device__ __forceinline__ void load(const GemmParams& p, buf_rsrc rs, int k_begin, int k_end, int kt, u32x4 (&r)[4]) const {
        const int prow = k_begin + kt * GEMM_BK + 4 * (threadIdx.x >> 4);
        const int hw = p.g_h_log2 + p.g_w_log2;
        const int n = prow >> hw, oy = (prow >> p.g_w_log2) & ((1 << p.g_h_log2) - 1), ox = prow & ((1 << p.g_w_log2) - 1);
        const int sy = oy * p.s_mul + dy;
        const bool row_ok = col_ok && (unsigned)sy < (unsigned)p.s_h;
        const unsigned rowbase = (unsigned)((n * p.s_h + sy) * p.s_w);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sx = (ox + i) * p.s_mul + dx;
            const bool ok = row_ok && prow + i < k_end && (unsigned)sx < (unsigned)p.s_w;
            const unsigned o = ((rowbase + (unsigned)sx) * (unsigned)p.ldb + lane_c) * 2u;
            r[i] = buf_load16(rs, ok ? o : BUF_OOB);
        }
    }
};
__device__ __forceinline__ void gemm_store_tn(char* tile, const u32x4 (&r)[4]) {
    const int t = threadIdx.x;
    const int cb = t & 15, mb = t >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = 8 * cb + j;                         // tile row = output row/column index
        const unsigned w0 = r[0][j >> 1], w1 = r[1][j >> 1], w2 = r[2][j >> 1], w3 = r[3][j >> 1];
        u32x2 o;                                            // one v_perm_b32 per dword (shift + and + or were three)
        if (j & 1) {                                        // high halves of the four m rows
            o.x = perm_b32(w1, w0, 0x07060302u);
            o.y = perm_b32(w3, w2, 0x07060302u);
        } else {
            o.x = perm_b32(w1, w0, 0x05040100u);
            o.y = perm_b32(w3, w2, 0x05040100u);
        }
        *reinterpret_cast<u32x2*>(tile + row * 128 + gemm_swz(row, mb >> 1) * 16 + (mb & 1) * 8) = o;
    }
}

// bias8: the 8 bias values of columns gn .. gn + 7 when the caller loaded them ONCE per tile (gemm.h's row pass: the same 8
// columns in every pass - loaded inside the pass they sit behind the previous pass's store, and a wait behind a store is
// `s_waitcnt vmcnt(0)`: one store round trip per pass, 8 per tile; the head's K = 256 products spent 2/3 of a tile there)
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_row8(const GemmParams& p, int gm, int gn, float* v, const float* bias8 = nullptr) {
    if (EPI != EPI_ATOMIC && EPI != EPI_DGELU && p.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias8 ? bias8[e] : p.bias[gn + e];
    }
    if (EPI == EPI_BF16) {
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = pack8(v);
    } else if (EPI == EPI_GELU) {
        if (p.C) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = pack8(v);
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = gelu_f(v[e]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C2) + (long)gm * p.ldc2 + gn) = pack8(g);
    } else if (EPI == EPI_RESID) {
        const float s = p.rowscale ? p.rowscale[p.rps_shift >= 0 ? gm >> p.rps_shift : gm / p.rows_per_sample] : 1.0f;
        const f32x4v* rp = reinterpret_cast<const f32x4v*>(p.resid + (long)gm * p.ldr + gn);
        f32x4v r0 = rp[0], r1 = rp[1];
        f32x4v o0 = {r0.x + v[0] * s, r0.y + v[1] * s, r0.z + v[2] * s, r0.w + v[3] * s};
        f32x4v o1 = {r1.x + v[4] * s, r1.y + v[5] * s, r1.z + v[6] * s, r1.w + v[7] * s};
        f32x4v* op = reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn);
        op[0] = o0;
        op[1] = o1;
    } else if (EPI == EPI_F32) {
        f32x4v* op = reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn);
        f32x4v o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        op[0] = o0;
        op[1] = o1;
    } else if (EPI == EPI_ATOMIC) {
        float* op = reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(op + e, v[e]);
    } else if (EPI == EPI_DGELU) {
        const u32x4 uw = *reinterpret_cast<const u32x4*>(p.aux + (long)gm * p.ldaux + gn);
        float u[8];
        unpack8(uw, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dgelu_f(u[e]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = pack8(v);
        if (p.C2) {                  // gelu(u) for the weight-gradient product that follows (the forward pass kept only u)
            float g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = u[e] * gelu_terms(u[e]).cdf;
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C2) + (long)gm * p.ldc2 + gn) = pack8(g);
        }
    }
}

// Persistent kernel: the grid is min(work items, 2 per CU); a workgroup walks its XCD's share of the (tile, split)
// items.  While the accumulators of one tile go through the epilogue, the first two k-tiles of the NEXT tile are already
// in flight (their registers are free by then), the epilogue's global stores drain under the next main loop, and the
// ~2.5 us launch-to-first-MFMA latency of a workgroup is paid once per CU slot instead of once per tile.
template <bool TN, int EPI, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
    const int m_static = p.M;                               // the work list was built for the static shape
    if (p.d_rows) {
        const int dyn = p.d_rows[0] * p.rows_mul;
        if (TN) p.K = dyn < p.K ? dyn : p.K;
        else p.M = dyn < p.M ? dyn : p.M;
    }
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;

    // NT with a device-side row count: only the LIVE row tiles are enumerated (the head runs with a static bound of
    // 2*26*B rows of which ~25 % exist; walking the dead tiles one by one cost ~12 % of the logits product)
    const int tiles_m = ((TN ? m_static : p.M) + GEMM_BM - 1) / GEMM_BM, tiles_n = (p.N + GEMM_BN - 1) / GEMM_BN;
    // Work items = tiles x splits, numbered so that consecutive ids share work: for NT the column tiles of one A
    // row-panel, for TN all output tiles of ONE contraction slice (they stream the same dY / X rows).  XCD x (the
    // hardware sends workgroup b to XCD b % 8) owns one contiguous range of ids, its workgroups take them round-robin,
    // so neighbours run at the same time on the same L2 and the re-reads are L2 hits instead of HBM traffic.
    const unsigned ntile = (unsigned)(tiles_m * tiles_n);
    // (NT: the contraction is cut too when k_per_split < K - EPI_ATOMIC products with few output tiles and a long K)
    const unsigned nt_splits = TN ? 1u : (unsigned)((p.K + p.k_per_split - 1) / p.k_per_split);
    const unsigned total = TN ? (unsigned)p.work_items : ntile * nt_splits, G = gridDim.x;
    const unsigned ng = G < 8u ? G : 8u;                                       // XCDs that received workgroups
    const unsigned xcd = blockIdx.x % ng, slot = blockIdx.x / ng;
    const unsigned nx = G / ng + (xcd < G % ng ? 1u : 0u);                     // workgroups on this XCD
    const unsigned q8 = total / ng, r8 = total % ng;
    const unsigned base_x = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned cnt_x = q8 + (xcd < r8 ? 1u : 0u);
    if (slot >= cnt_x) return;

    int m0, n0, k_begin, k_end, nk;
    auto decode = [&](unsigned item) {
        const unsigned lin = base_x + item;
        const unsigned tile = lin % ntile, split = lin / ntile;
        int tm, tn;
        if (p.m_fastest & 1) { tm = tile % tiles_m; tn = tile / tiles_m; }
        else { tn = tile % tiles_n; tm = tile / tiles_n; }
        m0 = tm * GEMM_BM;
        n0 = tn * GEMM_BN;
        k_begin = 0;
        k_end = p.K;
        if (TN || nt_splits > 1) {
            k_begin = split * p.k_per_split;
            k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
        }
        nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;
        if (!TN && m0 >= p.M) nk = 0;                       // only possible with a device-side row count
    };

    // Two register sets hold the k-tiles t+1 and t+2 while tile t is multiplied out of LDS: global loads are issued
    // TWO iterations ahead of the LDS write that consumes them (the CU keeps ~2 x 32 KiB per workgroup in flight).
    u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    NtLoader nta, ntb;
    TnLoader tna, tnb;
    GatherLoader gta;
    TnGatherLoader tgb;
    auto init_loaders = [&]() {
        if (TN) {
            tna.init(p.lda, m0, p.M);
            if (GATHER) tgb.init(p, n0, p.N);
            else tnb.init(p.ldb, n0, p.N);
        } else {
            if (GATHER) gta.init(p, m0);
            else nta.init(p.lda, m0, p.M);
            ntb.init(p.ldb, n0, p.N);
        }
    };
    // k-tile kt -> registers.  Tiles past the last one (the prefetch runs two ahead) get an empty descriptor: zeros.
    auto load = [&](int kt, u32x4 (&ra)[4], u32x4 (&rb)[4]) {
        const unsigned whole = kt < nk ? BUF_OOB : 0u;      // "whole buffer": every valid offset is below BUF_OOB
        if (TN) {
            tna.load(gemm_tn_rsrc(p.A, p.lda, k_begin, kt < nk ? k_end : k_begin, kt), ra);
            if (GATHER) tgb.load(p, make_rsrc(p.B, whole), k_begin, k_end, kt, rb);
            else tnb.load(gemm_tn_rsrc(p.B, p.ldb, k_begin, kt < nk ? k_end : k_begin, kt), rb);
        } else {
            if (GATHER) gta.load(p, make_rsrc(p.A, whole), kt, ra);
            else nta.load(make_rsrc(p.A + k_begin + kt * GEMM_BK, whole), ra);
            ntb.load(make_rsrc(p.B + k_begin + kt * GEMM_BK, whole), rb);
        }
    };
    // operand-A column sums (TN): kept per thread while the work item's k-tiles pass through the staging registers; only the
    // items of the first column tile (n0 == 0) carry them, so every row of A is summed exactly once
    float asum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool want_asum = false;
    auto store = [&](int stage, const u32x4 (&ra)[4], const u32x4 (&rb)[4]) {
        char* as = smem + stage * 2 * GEMM_STAGE_BYTES;
        char* bs = as + GEMM_STAGE_BYTES;
        if (TN) { gemm_store_tn(as, ra); gemm_store_tn(bs, rb); }
        else { gemm_store_nt(as, ra); gemm_store_nt(bs, rb); }
        if (TN && !GATHER && want_asum) {                    // this thread's 4 contraction rows x 8 columns of dY
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    asum[2 * jj] += bf_lo(ra[i][jj]);
                    asum[2 * jj + 1] += bf_hi(ra[i][jj]);
                }
        }
    };
    f32x16 acc[2][2];
    auto compute = [&](int stage) {
        const char* as = smem + stage * 2 * GEMM_STAGE_BYTES;
        const char* bs = as + GEMM_STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int slot16 = 2 * kk + (lane >> 5);
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 64 * wm + 32 * i + (lane & 31);
                a[i] = *reinterpret_cast<const bf16x8*>(as + row * 128 + gemm_swz(row, slot16) * 16);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 64 * wn + 32 * j + (lane & 31);
                b[j] = *reinterpret_cast<const bf16x8*>(bs + row * 128 + gemm_swz(row, slot16) * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16_bf16(a[i], b[j], acc[i][j]);
        }
    };
#ifdef CCD_GEMM_LAB     // timing ablations (tools/gemm_lab.py): bits of m_fastest switch phases off
    const int lab = p.m_fastest;
#define LAB_ON(bit) (!(lab & (bit)))
#else
#define LAB_ON(bit) true
#endif

#ifdef CCD_GEMM_LAB     // lab bit 64: per-phase cycle totals of wave 0 of each workgroup -> p.colsum (8 u64 per workgroup)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define LAB_STAMP(i) if (lab & 64) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tprev; tprev = tn_; }
#else
#define LAB_STAMP(i)
#endif
    // column statistics (bias gradients / BatchNorm sums): with a single column tile (N <= 128: every convolution of the
    // segmentation head) all of a workgroup's tiles cover the same columns, so the sums stay in registers across tiles
    // and are published once - same-address fp32 atomics cost ~0.3 ns each chip-wide and there were 4 M of them per step
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float csq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned item = slot;
    decode(item);
    want_asum = TN && p.colsum_a != nullptr && n0 == 0;
    init_loaders();
    load(0, ra0, rb0);
    load(1, ra1, rb1);
    while (true) {
        store(0, ra0, rb0);
        load(2, ra0, rb0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        LAB_STAMP(0)
        __syncthreads();
        LAB_STAMP(1)
        // Branch-free steady state: LDS writes and loads past the last k-tile move zeros, an odd tile count multiplies
        // one all-zero stage.  (A multi-exit loop made the register allocator shuttle all 64 accumulator registers
        // between two homes every iteration - v_mov chains that wait on the MFMA results.)
        for (int kt = 0; kt < nk; kt += 2) {
            if (LAB_ON(8)) compute(0);                      // tile kt (even) lives in stage 0
            LAB_STAMP(2)
            if (LAB_ON(16)) store(1, ra1, rb1);             // tile kt + 1
            LAB_STAMP(3)
            if (LAB_ON(32)) load(kt + 3, ra1, rb1);
            LAB_STAMP(4)
            __syncthreads();
            LAB_STAMP(1)
            if (LAB_ON(8)) compute(1);                      // tile kt + 1
            LAB_STAMP(2)
            if (LAB_ON(16)) store(0, ra0, rb0);             // tile kt + 2
            LAB_STAMP(3)
            if (LAB_ON(32)) load(kt + 4, ra0, rb0);
            LAB_STAMP(4)
            __syncthreads();
            LAB_STAMP(1)
        }
        if (TN && !GATHER && want_asum) {                    // (wave-uniform) 16 contraction blocks x 128 columns -> one atomic per column
            float* red = reinterpret_cast<float*>(smem);
            const int cb = t & 15, mb = t >> 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[mb * 128 + 8 * cb + j] = asum[j]; asum[j] = 0.f; }
            __syncthreads();
            if (t < 128 && m0 + t < p.M) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) a += red[q * 128 + t];
                atomicAdd(p.colsum_a + m0 + t, a);
            }
            __syncthreads();
        }
        // ---- next work item: its first two k-tiles fly while this tile's accumulators are written out
        const int em0 = m0, en0 = n0;
        const unsigned next = item + nx;
        const bool has_next = next < cnt_x;
        if (has_next) {
            decode(next);
            want_asum = TN && p.colsum_a != nullptr && n0 == 0;
            init_loaders();
            load(0, ra0, rb0);
            load(1, ra1, rb1);
        }
        LAB_STAMP(5)
        if (LAB_ON(4) || p.alpha == 12345.f) {
        // ---- epilogue: accumulators -> fp32 LDS tile -> row-contiguous 8-wide global accesses
        float* cs = reinterpret_cast<float*>(smem);
        if (p.alpha != 1.0f) {                               // wave-uniform: 64 multiplies only when asked for
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.alpha;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int col = 64 * wn + 32 * j + (lane & 31);
                    cs[row * GEMM_CS_LD + col] = acc[i][j][r];
                }
        LAB_STAMP(6)
        __syncthreads();
        LAB_STAMP(1)
        if (EPI == EPI_ATOMIC) {     // 64 consecutive lanes -> 64 consecutive floats of one row: one 256-B atomic burst
            const int col = t & 127, gn = en0 + col;
            if (gn < p.N) {
#pragma unroll 4
                for (int row = t >> 7; row < GEMM_BM; row += 2) {
                    const int gm = em0 + row;
                    if (gm < p.M) atomicAdd(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn, cs[row * GEMM_CS_LD + col]);
                }
            }
        } else {
            const bool want_stats = p.colsum != nullptr || p.colsumsq != nullptr;
            const bool publish = tiles_n > 1 || !has_next;   // defer while the next tile has the same columns
            float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (EPI != EPI_DGELU && p.bias && en0 + (t & 15) * 8 < p.N) {
                const f32x4v b0 = *reinterpret_cast<const f32x4v*>(p.bias + en0 + (t & 15) * 8);
                const f32x4v b1 = *reinterpret_cast<const f32x4v*>(p.bias + en0 + (t & 15) * 8 + 4);
                bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
                bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
#pragma unroll
                for (int e = 0; e < 8; ++e) needed_here(bias8[e]);
            }
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {       // unrolled: the 16 LDS reads go out first, the stores stream
                const int row = pass * 16 + (t >> 4), col = (t & 15) * 8;
                const int gm = em0 + row, gn = en0 + col;
                if (gm < p.M && gn < p.N) {
                    float v[8];
                    const f32x4v c0 = *reinterpret_cast<const f32x4v*>(cs + row * GEMM_CS_LD + col);
                    const f32x4v c1 = *reinterpret_cast<const f32x4v*>(cs + row * GEMM_CS_LD + col + 4);
                    v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
                    v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
                    int gm_out = gm;
                    if (GATHER && p.c_map) {          // transposed-conv parity class: scatter to the 2x upsampled grid
                        const int hw = p.g_h_log2 + p.g_w_log2;
                        const int n = gm >> hw, oy = (gm >> p.g_w_log2) & ((1 << p.g_h_log2) - 1), ox = gm & ((1 << p.g_w_log2) - 1);
                        gm_out = (((n << (p.g_h_log2 + 1)) + 2 * oy + p.c_py) << (p.g_w_log2 + 1)) + 2 * ox + p.c_px;
                    }
                    if (LAB_ON(2)) gemm_epilogue_row8<EPI>(p, gm_out, gn, v, bias8);
                    if ((EPI == EPI_DGELU || EPI == EPI_BF16) && want_stats) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { csum[e] += v[e]; csq[e] += v[e] * v[e]; }
                    }
                }
            }
            if (EPI == EPI_BF16 && p.colsumsq && publish) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 8; ++e) { cs[(t >> 4) * GEMM_BN + (t & 15) * 8 + e] = csq[e]; csq[e] = 0.f; }
                __syncthreads();
                if (t < GEMM_BN && en0 + t < p.N) {
                    float a = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) a += cs[q * GEMM_BN + t];
                    atomicAdd(p.colsumsq + en0 + t, a);
                }
            }
            if ((EPI == EPI_DGELU || EPI == EPI_BF16) && p.colsum && publish && LAB_ON(64)) {   // block-wide column sums -> one atomic per column
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 8; ++e) { cs[(t >> 4) * GEMM_BN + (t & 15) * 8 + e] = csum[e]; csum[e] = 0.f; }
                __syncthreads();
                if (t < GEMM_BN && en0 + t < p.N) {
                    float a = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) a += cs[q * GEMM_BN + t];
                    atomicAdd(p.colsum + en0 + t, a);
                }
            }
        }
        }
        LAB_STAMP(7)
        if (!has_next) break;
        item = next;
        __syncthreads();                                    // the staging tile is read out: LDS belongs to the stages again
        LAB_STAMP(1)
    }
#ifdef CCD_GEMM_LAB
    if ((lab & 64) && p.colsum && t == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(p.colsum)[blockIdx.x * 8 + i] = ph[i];
#endif
}

}  // namespace ccd
