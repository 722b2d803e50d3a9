// rowgemm16.h - the row-owner LayerNorm-backward product (rowgemm.h, RG_LNBWD) re-cut so that TWO workgroups share a CU:
// a wave owns 16 rows (v_mfma_f32_16x16x32_bf16: E/4 accumulator registers instead of E/2), a workgroup is 4 waves = 64
// rows with its own 4-slot ring of [128 output columns][64 k] weight pieces (16 KiB each), and two such workgroups -
// independent, each with its own barriers - are resident per CU (80 KiB of LDS, <= 256 registers per wave).
// Why: rowgemm.h is the SUM of an MFMA-bound and an HBM-bound phase - with one wave per SIMD nothing hides the epilogue's
// loads, and its stores have to drain before the next ring wait can pass (gfx950 counts loads and stores on one vmcnt and
// returns them out of order with respect to each other).  Two independent workgroups per CU are in different phases: one's
// epilogue (690 KB of row traffic per 128 rows) runs under the other's products.  The price is LDS bandwidth: every wave
// still reads every weight piece, now for 16 rows instead of 32 - the product phase is LDS-bound (128 B/clk/CU) at roughly
// the MFMA rate, which the measured 45 % MFMA efficiency of rowgemm.h's product phase leaves room for.
//
// Layouts.  out^T[n][row] tile = W piece rows (A operand from LDS: lane (i = l & 15, g = l >> 4) reads n-row 16 nt + i,
// k = 32 s + 8 g .. + 7) x activation rows (B operand straight from HBM: lane (row = l & 15, g) loads 16 bytes at
// A[row][64 b + 32 s + 8 g]); accumulator register r of tile nt = column 16 nt + 4 g + r of row l & 15.
// VMEM bookkeeping as in rowgemm.h: every piece window issues exactly KT = 4 ring requests and 2 other loads (the block's
// two activation loads in its first window, two zero-size dummies in the others), so "my quarter of piece q has landed"
// is vmcnt <= (AHEAD - 1) * 6.
#pragma once

namespace ccd {

constexpr int RG16_THREADS = 256, RG16_BM = 64, RG16_NSLOT = 4, RG16_PIECE = 128 * 64 * 2, RG16_SCRATCH = 2048;
__host__ __device__ inline int rg16_smem_bytes(int E) { return RG16_NSLOT * RG16_PIECE + 4 * RG16_SCRATCH + 4 * E * 4; }

// fragment k of a piece: MFMA k = (k-step k / 8, n-tile k % 8); the swizzle of odd n-tiles differs in one address bit
struct Rg16Map {
    static constexpr int reg(int k) { return 2 * (k / 8) + (k & 1); }
    static constexpr int off(int k) { return (k % 8) * 2048; }
};
// v[4 tl + r] = this lane's row, column 4 g + r of 16-column tile tl (four consecutive tiles): column sums over the 16 rows
// of the wave -> dst[64] (LDS).  After the four halving steps lane i of each 16-lane group owns tile i >> 2, column i & 3:
// every one of the 64 lanes adds ONE distinct column - no branch, no address conflict.
__device__ __forceinline__ void rg16_colsum16(const float (&v)[16], float* dst, int lane) {
    const int i = lane & 15;
    float a8[8], a4[4], a2[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) a8[j] = rg_fold<1>(v[2 * j], v[2 * j + 1], (i & 1) != 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) a4[j] = rg_fold<2>(a8[2 * j], a8[2 * j + 1], (i & 2) != 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) a2[j] = rg_fold<4>(a4[2 * j], a4[2 * j + 1], (i & 4) != 0);
    const float tot = rg_fold<8>(a2[0], a2[1], (i & 8) != 0);
    atomicAdd(dst + 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3), tot);
}

template <int E, int R, int DEPTH = 6>
__global__ __launch_bounds__(RG16_THREADS, 2) void rowgemm16_lnbwd_kernel(RowGemmParams p) {
    constexpr int KT = 4;                  // ring requests (1 KiB wave instructions) per wave and piece
    constexpr int PPB = E / 128;           // pieces per 64-wide k-block
    constexpr int NT = E / 16;             // 16-column tiles of a row
    constexpr int AHEAD = RG16_NSLOT - 1;
    constexpr int NSTEP = 16;              // MFMAs per piece: 8 n-tiles x 2 k-steps
    constexpr int WIN_VM = KT + 2;
    static_assert(E % 128 == 0 && AHEAD >= 2, "pieces are 128 output columns");
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, g4 = lane >> 4, li = lane & 15;
    const int w = uniform_i32(t >> 6);
    char* scratch = smem + RG16_NSLOT * RG16_PIECE + w * RG16_SCRATCH;
    float* vga = reinterpret_cast<float*>(smem + RG16_NSLOT * RG16_PIECE + 4 * RG16_SCRATCH);
    float* cs = vga + E;                   // [3][E]: dgamma, dbeta, dbias of this workgroup
    for (int i = t; i < E; i += RG16_THREADS) { vga[i] = p.gamma[i]; cs[i] = 0.f; cs[E + i] = 0.f; cs[2 * E + i] = 0.f; }
    __syncthreads();

    const int NB = p.K / 64, NP = PPB * NB;                   // k-blocks, pieces per row tile (NB % R == 0)
    const int tiles = (p.M + RG16_BM - 1) / RG16_BM, G = gridDim.x;

    // ---- weight ring: piece (blk, c) = W rows 128 c + 32 i + 8 w + (0..7), k = 64 blk ..; [128 rows][128 B] image
    const int dr = lane >> 3, dp = lane & 7, drow = 8 * w + dr;
    const unsigned req_lane = (unsigned)(2 * drow) * (unsigned)p.ldw + (unsigned)((dp ^ mlp_swz(drow)) * 16);
    const long req_step = 64 * p.ldw;                         // bytes between the 32-row groups of a piece
    int slot_i = 0, slot_c = 0, pos_blk = 0, pos_c = 0;
    const char* req_base = nullptr;
    char* req_lds = nullptr;
    auto issue_prepare = [&]() {
        req_base = reinterpret_cast<const char*>(p.W) + ((long)(128 * pos_c) * p.ldw + 64 * pos_blk) * 2;
        req_lds = smem + slot_i * RG16_PIECE + w * 1024;
        slot_i = slot_i + 1 == RG16_NSLOT ? 0 : slot_i + 1;
        if (++pos_c == PPB) { pos_c = 0; pos_blk = pos_blk + 1 == NB ? 0 : pos_blk + 1; }
    };
    auto issue_one = [&](int i) { glds16(req_base + i * req_step + req_lane, req_lds + 4096 * i); };
    const unsigned smem_addr = lds_addr_of(smem);
    auto acquire = [&]() -> unsigned {
        glds_wait<(AHEAD - 1) * WIN_VM>();
        lds_barrier();
        issue_prepare();
        const unsigned sb = smem_addr + (unsigned)(slot_c * RG16_PIECE);
        slot_c = slot_c + 1 == RG16_NSLOT ? 0 : slot_c + 1;
        return sb;
    };
    (void)NP;
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
        issue_prepare();
#pragma unroll
        for (int i = 0; i < KT; ++i) issue_one(i);
    }
    // fragment addresses inside a piece: [k-step s][n-tile parity]
    unsigned offf[4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int par = 0; par < 2; ++par)
            offf[2 * s + par] = (unsigned)(li * 128 + (((4 * s + g4) ^ (((li >> 1) ^ par) & 7)) * 16));

    const buf_rsrc rs_a = make_rsrc(p.A, (unsigned)((((long)p.M - 1) * p.lda + p.K) * 2));
    const buf_rsrc rs_x = make_rsrc(p.x, (unsigned)((((long)p.M - 1) * p.ldx + E) * 4));
    const buf_rsrc rs_g = make_rsrc(p.g, (unsigned)((((long)p.M - 1) * p.ldg + E) * 4));
    const buf_rsrc rs_b = make_rsrc(p.gb, p.gb ? (unsigned)((((long)p.M - 1) * p.ld_gb + E) * 2) : 0u);
    struct LaneOff {
        int lane, g4, li, dr, dp;
        __device__ __forceinline__ explicit LaneOff(int t) {
            lane = opaque_vgpr(t) & 63; g4 = lane >> 4; li = lane & 15; dr = lane >> 3; dp = lane & 7;
        }
        __device__ __forceinline__ unsigned frag(long ld, int elt, int per_g) const { return (unsigned)((li * ld + per_g * g4) * elt); }
        __device__ __forceinline__ unsigned rows8(long ld, int elt) const { return (unsigned)(dr * ld * elt + dp * 16); }
        __device__ __forceinline__ unsigned scr_rd(int i) const { return (unsigned)((dr + 8 * i) * 128 + ((dp ^ dr) * 16)); }
        __device__ __forceinline__ unsigned scr_wr(int slot16) const { return (unsigned)(li * 128 + ((slot16 ^ (li & 7)) * 16)); }
    };
    // this lane's row of A: k-block b of row tile tt, k-step s = the B operand (k = 64 b + 32 s + 8 g .. + 7)
    const unsigned lo_a = (unsigned)((li * p.lda + 8 * g4) * 2);
    u32x4 ab[R][2];
    auto load_a = [&](u32x4& dst, int tt, int b, int s) {
        dst = buf_load16(rs_a, lo_a, (unsigned)(tt * RG16_BM + 16 * w) * (unsigned)(p.lda * 2) + (unsigned)(128 * b + 64 * s));
    };
    unsigned vm_pad = 0u;                                      // sink of the padding loads (stays live to the kernel's end)
    auto dummy_load = [&]() { vmem_pad_load(vm_pad); };        // keeps the window's VMEM count at KT + 2 (see the header)
    if ((int)blockIdx.x < tiles) {
#pragma unroll
        for (int b = 0; b < R - 1; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) load_a(ab[b][s], blockIdx.x, b, s);
    }
    if constexpr (2 * (R - 1) < 2 * (AHEAD - 1)) glds_wait_all();
    const float inv_e = 1.0f / (float)E;
    // (lab: the second workgroup of every CU starts late - two workgroups launched together run the same phase at the same time)
    if (p.lab > 0 && blockIdx.x >= gridDim.x / 2) wave_sleep(p.lab);

    for (int tile = blockIdx.x; tile < tiles; tile += G) {
        const int m0 = tile * RG16_BM, r0 = m0 + 16 * w;
        const int row = r0 + li, grow = row < p.M ? row : p.M - 1;
        const float mu = p.mean[grow], rs = p.rstd[grow];
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int grp = 0; grp < NB / R; ++grp) {
            mlp_static_for<0, R>([&](auto I) {
                constexpr int i = decltype(I)::value, sp = (i + R - 1) % R;
                int tb = grp * R + i + R - 1, tt = tile;
                if (tb >= NB) { tb -= NB; tt += G; }
                mlp_static_for<0, PPB>([&](auto C) {
                    constexpr int c = decltype(C)::value;
                    const unsigned sb = acquire();
                    const unsigned areg[4] = {sb + offf[0], sb + offf[1], sb + offf[2], sb + offf[3]};
                    mlp_product<NSTEP, DEPTH, Rg16Map, MlpNoExtra>(
                        areg,
                        [&](auto K, const bf16x8& a) {
                            constexpr int k = decltype(K)::value, nt = 8 * c + k % 8;
                            acc[nt] = mfma_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, ab[i][k / 8]), acc[nt]);
                        },
                        [&](auto K) {
                            constexpr int k = decltype(K)::value;
                            if constexpr (k % 4 == 1) issue_one(k / 4);
                            if constexpr (k == 0 || k == 2) {
                                if constexpr (c == 0) load_a(ab[sp][k / 2], tt, tb, k / 2);
                                else dummy_load();
                            }
                        });
                });
            });
        }
        // ---- epilogue: a row is complete inside its four lanes (l & 15 fixed); acc[nt][r] = column 16 nt + 4 g + r.
        // Pass A: the two row means, the column sums of dy * xhat and dy; (dy * gamma, xhat) packed back into the accumulator
        const unsigned so_x = (unsigned)r0 * (unsigned)(p.ldx * 4);
        float s1 = 0.f, sq = 0.f;
        {
            const unsigned lo_x = LaneOff(t).frag(p.ldx, 4, 4);
            constexpr int PA = 6;
            u32x4 xb[PA];
            float vg[16], vb[16];
#pragma unroll
            for (int nt = 0; nt < PA - 1 && nt < NT; ++nt) xb[nt] = buf_load16(rs_x, lo_x, so_x + 64 * nt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt + PA - 1 < NT) xb[(nt + PA - 1) % PA] = buf_load16(rs_x, lo_x, so_x + 64 * (nt + PA - 1));
                const f32x4v x = __builtin_bit_cast(f32x4v, xb[nt % PA]);
                const f32x4v ga = *reinterpret_cast<const f32x4v*>(vga + 16 * nt + 4 * g4);
                const float xx[4] = {x.x, x.y, x.z, x.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dy = acc[nt][e], xh = (xx[e] - mu) * rs, dg = dy * gg[e];
                    acc[nt][e] = opaque_f32(__builtin_bit_cast(float, rg_pack(dg, xh)));
                    s1 += dg;
                    sq = fmaf(dg, xh, sq);
                    vg[4 * (nt & 3) + e] = dy * xh;
                    vb[4 * (nt & 3) + e] = dy;
                }
                if (nt % 4 == 3) {
                    rg16_colsum16(vg, cs + 16 * (nt - 3), lane);
                    rg16_colsum16(vb, cs + E + 16 * (nt - 3), lane);
                    CCD_SCHED_FENCE();
                    asm volatile("" ::: "memory");
                }
            }
        }
        s1 += shfl_xor(s1, 16); s1 += shfl_xor(s1, 32);
        sq += shfl_xor(sq, 16); sq += shfl_xor(sq, 32);
        const float c1 = rs * s1 * inv_e, c2 = rs * sq * inv_e;
        float sc = 1.0f;
        if (p.gb && p.rowscale) sc = p.rowscale[grow / p.rows_per_sample];
        // Pass B: dx, g (fp32, two tiles = 128 bytes of a row at a time) and gb (bf16, four tiles) through the scratch image
        {
            const LaneOff lo(t);
            const unsigned lo_gl = lo.frag(p.ldg, 4, 4), so_g = (unsigned)r0 * (unsigned)(p.ldg * 4);
            const unsigned lo_o = lo.rows8(p.ldg, 4), lo_n = lo.rows8(p.ld_gb, 2);
            constexpr int PB = 6;
            u32x4 gbuf[PB];
            auto load_g = [&](int nt) {
                gbuf[nt % PB] = u32x4{0u, 0u, 0u, 0u};
                if (p.accumulate) gbuf[nt % PB] = buf_load16(rs_g, lo_gl, so_g + 64 * nt);
            };
#pragma unroll
            for (int nt = 0; nt < PB - 1 && nt < NT; ++nt) load_g(nt);
#pragma unroll
            for (int nq = 0; nq < NT / 4; ++nq) {
                u32x2 ypk[4];
                float vbi[16];
#pragma unroll
                for (int np = 0; np < 2; ++np) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int nt = 4 * nq + 2 * np + tt;
                        if (nt + PB - 1 < NT) load_g(nt + PB - 1);
                        const f32x4v go = __builtin_bit_cast(f32x4v, gbuf[nt % PB]);
                        const float oo[4] = {go.x, go.y, go.z, go.w};
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float pf = acc[nt][e];
                            const unsigned pk = __builtin_bit_cast(unsigned, pf);
                            v[e] = oo[e] + fmaf(-c2, rg_unpack_xh(pk), fmaf(rg_unpack_dg(pk), rs, -c1));
                        }
                        *reinterpret_cast<f32x4v*>(scratch + lo.scr_wr(4 * tt + lo.g4)) = f32x4v{v[0], v[1], v[2], v[3]};
                        if (p.gb) {
                            ypk[2 * np + tt].x = pack_bf2(v[0] * sc, v[1] * sc);
                            ypk[2 * np + tt].y = pack_bf2(v[2] * sc, v[3] * sc);
                            const int q4 = 4 * (2 * np + tt);
                            vbi[q4] = bf_lo(ypk[2 * np + tt].x); vbi[q4 + 1] = bf_hi(ypk[2 * np + tt].x);
                            vbi[q4 + 2] = bf_lo(ypk[2 * np + tt].y); vbi[q4 + 3] = bf_hi(ypk[2 * np + tt].y);
                        }
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                        buf_store16(rs_g, lo_o, (unsigned)(r0 + 8 * i) * (unsigned)(p.ldg * 4) + 128 * (2 * nq + np), o);
                    }
                    wave_lds_fence();
                    CCD_SCHED_FENCE();
                    asm volatile("" ::: "memory");
                }
                if (p.gb) {
                    if (p.dbias) rg16_colsum16(vbi, cs + 2 * E + 64 * nq, lane);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<u32x2*>(scratch + lo.scr_wr(2 * q + (lo.g4 >> 1)) + 8 * (lo.g4 & 1)) = ypk[q];
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const u32x4 o = *reinterpret_cast<const u32x4*>(scratch + lo.scr_rd(i));
                        buf_store16(rs_b, lo_n, (unsigned)(r0 + 8 * i) * (unsigned)(p.ld_gb * 2) + 128 * nq, o);
                    }
                    wave_lds_fence();
                }
            }
        }
    }
    glds_wait_all();                       // requested pieces that no tile consumed must not outlive the workgroup's LDS
    __syncthreads();
    for (int i = t; i < E; i += RG16_THREADS) {
        atomicAdd(p.dgamma + i, cs[i]);
        atomicAdd(p.dbeta + i, cs[E + i]);
        if (p.dbias) atomicAdd(p.dbias + i, cs[2 * E + i]);
    }
    if (vm_pad == 0x5EEDF00Du) p.dgamma[0] = 0.f;             // (never true: the padding loads return zeros)
}

}  // namespace ccd
