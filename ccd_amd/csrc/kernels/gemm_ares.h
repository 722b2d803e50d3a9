// gemm_ares.h - "A-resident" NT GEMM for the transformer's short-K products (K <= 384: qkv, proj, fc1 forward;
// dGELU and d_attn backward):  C[M,N] = A[M,K] . B[N,K]^T + fused epilogue.
//
// The generic 128x128 tile kernel (gemm.h) re-stages the A panel for every column tile and pays a pipeline fill and
// an LDS-staged epilogue per 6 k-steps; at K = 384 its instruction issue is dominated by that fixed cost (PMC: 14
// VALU per MFMA).  Here ONE persistent workgroup (8 waves, 1 per CU) keeps a whole 128-row A panel (128 x K bf16 =
// 96 KiB at K = 384) in LDS and streams only the weight tiles (L2-resident) through a 2 x 32 KiB ring while it walks
// ALL column tiles of that panel; the next panel is prefetched into registers under the last column tile.
//   * A is read from HBM exactly once and written to LDS once per panel (was: once per column tile).
//   * accumulators are produced TRANSPOSED (mfma(B-frag, A-frag)), so a lane owns 4 consecutive output columns of one
//     row: the epilogue stores 8-byte (bf16) / 16-byte (fp32) vectors straight from registers - no LDS round trip.
// LDS images are the same swizzled [rows][64 k] 16-KiB chunks as in gemm.h.
#pragma once

namespace ccd {

constexpr int ARES_THREADS = 512;
constexpr int ARES_MAX_KC = 6;                                   // K <= 384
constexpr int ARES_SMEM_BYTES = (ARES_MAX_KC + 4) * 16384;      // A panel (6 chunks) + 2 stages x 2 chunks of B

// stores 4 consecutive columns of one row and returns the values it stored (before bf16 rounding)
template <int EPI>
__device__ __forceinline__ f32x4v ares_store4(const GemmParams& p, int gm, int gn, float v0, float v1, float v2, float v3,
                                              float scale) {
    if (EPI != EPI_DGELU && p.bias) {
        const f32x4v b = *reinterpret_cast<const f32x4v*>(p.bias + gn);
        v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
    }
    if (EPI == EPI_BF16) {
        u32x2 o;
        o.x = pack_bf2(v0, v1);
        o.y = pack_bf2(v2, v3);
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = o;
    } else if (EPI == EPI_GELU) {
        u32x2 o;
        if (p.C) {
            o.x = pack_bf2(v0, v1);
            o.y = pack_bf2(v2, v3);
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = o;
        }
        o.x = pack_bf2(gelu_f(v0), gelu_f(v1));
        o.y = pack_bf2(gelu_f(v2), gelu_f(v3));
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C2) + (long)gm * p.ldc2 + gn) = o;
    } else if (EPI == EPI_RESID) {
        const f32x4v r = *reinterpret_cast<const f32x4v*>(p.resid + (long)gm * p.ldr + gn);
        const f32x4v o = {r.x + v0 * scale, r.y + v1 * scale, r.z + v2 * scale, r.w + v3 * scale};
        *reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn) = o;
    } else if (EPI == EPI_DGELU) {
        const u32x2 uw = *reinterpret_cast<const u32x2*>(p.aux + (long)gm * p.ldaux + gn);
        v0 *= dgelu_f(bf_lo(uw.x)); v1 *= dgelu_f(bf_hi(uw.x)); v2 *= dgelu_f(bf_lo(uw.y)); v3 *= dgelu_f(bf_hi(uw.y));
        u32x2 o;
        o.x = pack_bf2(v0, v1);
        o.y = pack_bf2(v2, v3);
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn) = o;
    }
    const f32x4v out = {v0, v1, v2, v3};
    return out;
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_ares_kernel(GemmParams p) {
    char* smem = dynamic_smem();
    char* a_img = smem;                                         // [KC][128 rows][64 k]
    char* b_ring = smem + ARES_MAX_KC * 16384;                  // [2 stages][2 chunks][128 n][64 k]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hf = lane >> 5, lq = lane & 31;
    const int wm = w >> 2, wn = w & 3;                           // wave tile: rows 64*wm .. +63, columns 32*wn .. +31
    const int KC = p.K / 64;                                     // even (host guarantees K % 128 == 0)
    const int steps = KC / 2;                                    // a ring stage carries two 64-wide k chunks
    const int tiles_m = (p.M + 127) / 128, tiles_n = (p.N + 127) / 128;
    const int total_q = tiles_n * steps;
    const int arow = t >> 3, aslot = t & 7;                      // this thread's 16-B chunk inside a [64 rows][8 slots] half

    // ---- everything below is loop invariant: LDS byte offsets of this lane's fragments and staging slots
    int a_off[2][4], b_off[4], wr_off[2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int brow = 32 * wn + lq;
        b_off[kk] = brow * 128 + gemm_swz(brow, 2 * kk + hf) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 64 * wm + 32 * i + lq;
            a_off[i][kk] = row * 128 + gemm_swz(row, 2 * kk + hf) * 16;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) wr_off[h] = (arow + 64 * h) * 128 + gemm_swz(arow + 64 * h, aslot) * 16;

    // All global loads are unconditional: out-of-range rows are clamped to the last valid row (their products land in
    // accumulator rows/columns the epilogue never stores), so the compiler can count vmcnt exactly.
    u32x4 ar[2 * ARES_MAX_KC];
    auto fetch_a = [&](int tm) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int row = tm * 128 + arow + 64 * h;
            row = row < p.M ? row : p.M - 1;
            const bf16_t* src = p.A + (long)row * p.lda + aslot * 8;
#pragma unroll
            for (int kc = 0; kc < ARES_MAX_KC; ++kc)
                ar[2 * kc + h] = *reinterpret_cast<const u32x4*>(src + (kc < KC ? kc : 0) * 64);
        }
    };
    auto commit_a = [&]() {
#pragma unroll
        for (int kc = 0; kc < ARES_MAX_KC; ++kc)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (kc < KC) *reinterpret_cast<u32x4*>(a_img + kc * 16384 + wr_off[h]) = ar[2 * kc + h];
    };
    // B stream cursor: item = (column tile, step); thread rows n = 128*tn + arow (+64)
    const bf16_t* pb[2];
    int cur_tn = 0, cur_st = 0;
    auto reset_b = [&]() {
        cur_tn = 0;
        cur_st = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int n = arow + 64 * h;
            n = n < p.N ? n : p.N - 1;
            pb[h] = p.B + (long)n * p.ldb + aslot * 8;
        }
    };
    const int dbg = p.m_fastest;                                 // timing ablations only (bits 2: stores 4: loads 8: LDS writes 16: barrier)
    auto fetch_b = [&](u32x4 (&br)[4]) {                         // loads the current item, then advances (clamped at the end)
        if (dbg & 4) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            br[h] = *reinterpret_cast<const u32x4*>(pb[h]);          // chunk 2*st
            br[2 + h] = *reinterpret_cast<const u32x4*>(pb[h] + 64);  // chunk 2*st + 1
        }
        if (cur_st + 1 < steps) {
            ++cur_st;
            pb[0] += 128;
            pb[1] += 128;
        } else if (cur_tn + 1 < tiles_n) {
            ++cur_tn;
            cur_st = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int n = cur_tn * 128 + arow + 64 * h;
                n = n < p.N ? n : p.N - 1;
                pb[h] = p.B + (long)n * p.ldb + aslot * 8;
            }
        }
    };
    auto commit_b = [&](int stage, const u32x4 (&br)[4]) {
        if (dbg & 8) return;
        char* dst = b_ring + stage * 32768;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<u32x4*>(dst + wr_off[h]) = br[h];
            *reinterpret_cast<u32x4*>(dst + 16384 + wr_off[h]) = br[2 + h];
        }
    };

    f32x16 acc[2];
    auto compute = [&](int stage, int st) {                      // 2 chunks x 4 k-steps; fragments fetched one step ahead
        const char* ai = a_img + st * 32768;
        const char* bi = b_ring + stage * 32768;
        bf16x8 bf = *reinterpret_cast<const bf16x8*>(bi + b_off[0]);
        bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ai + a_off[0][0]);
        bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ai + a_off[1][0]);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            bf16x8 nbf = bf, na0 = a0, na1 = a1;
            if (s8 < 7) {
                const int c = ((s8 + 1) >> 2) * 16384, kk = (s8 + 1) & 3;
                nbf = *reinterpret_cast<const bf16x8*>(bi + c + b_off[kk]);
                na0 = *reinterpret_cast<const bf16x8*>(ai + c + a_off[0][kk]);
                na1 = *reinterpret_cast<const bf16x8*>(ai + c + a_off[1][kk]);
            }
            acc[0] = mfma_32x32x16_bf16(bf, a0, acc[0]);          // D[n][m]: transposed accumulator
            acc[1] = mfma_32x32x16_bf16(bf, a1, acc[1]);
            bf = nbf; a0 = na0; a1 = na1;
        }
    };

    int tm = blockIdx.x;
    if (tm >= tiles_m) return;
    fetch_a(tm);
    for (; tm < tiles_m; tm += gridDim.x) {
        u32x4 br0[4], br1[4];
        __syncthreads();                                         // previous panel fully consumed
        commit_a();
        reset_b();
        fetch_b(br0);                                            // item 0
        commit_b(0, br0);
        fetch_b(br1);                                            // item 1 -> set 1
        fetch_b(br0);                                            // item 2 -> set 0
        __syncthreads();
        float row_scale[2] = {1.0f, 1.0f};                      // DropPath scale of this lane's two rows
        if (EPI == EPI_RESID && p.rowscale) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int gm = tm * 128 + 64 * wm + 32 * i + lq;
                if (gm < p.M) row_scale[i] = p.rowscale[gm / p.rows_per_sample];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        int tn = 0, st = 0;
        const bool more_panels = tm + (int)gridDim.x < tiles_m;
        for (int q = 0; q < total_q; ++q) {
            // prefetch of the NEXT panel rides under the last column tile of this one
            if (more_panels && tn == tiles_n - 1 && st == 0) fetch_a(tm + gridDim.x);
            if ((q & 1) == 0) {
                compute(0, st);
                commit_b(1, br1);                                // item q + 1
                fetch_b(br1);                                    // item q + 3
            } else {
                compute(1, st);
                commit_b(0, br0);
                fetch_b(br0);
            }
            if (!(dbg & 16)) __syncthreads();
            if (++st == steps) {
                // ---- epilogue of column tile tn, straight from registers
                f32x4v csum[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) { csum[g].x = 0.f; csum[g].y = 0.f; csum[g].z = 0.f; csum[g].w = 0.f; }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int gm = tm * 128 + 64 * wm + 32 * i + lq;
                    if (gm < p.M && !(p.m_fastest & 2)) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int gn = tn * 128 + 32 * wn + 8 * g + 4 * hf;
                            if (gn < p.N)
                                csum[g] += ares_store4<EPI>(p, gm, gn, acc[i][4 * g] * p.alpha, acc[i][4 * g + 1] * p.alpha,
                                                            acc[i][4 * g + 2] * p.alpha, acc[i][4 * g + 3] * p.alpha,
                                                            row_scale[i]);
                        }
                    }
                }
                if ((EPI == EPI_DGELU || EPI == EPI_BF16) && p.colsum) {   // bias gradient: sum over the wave's 64 rows
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int msk = 16; msk >= 1; msk >>= 1) {
                            csum[g].x += shfl_xor(csum[g].x, msk); csum[g].y += shfl_xor(csum[g].y, msk);
                            csum[g].z += shfl_xor(csum[g].z, msk); csum[g].w += shfl_xor(csum[g].w, msk);
                        }
                        const int gn = tn * 128 + 32 * wn + 8 * g + 4 * hf;
                        if (lq == 0 && gn < p.N) {
                            atomicAdd(p.colsum + gn, csum[g].x); atomicAdd(p.colsum + gn + 1, csum[g].y);
                            atomicAdd(p.colsum + gn + 2, csum[g].z); atomicAdd(p.colsum + gn + 3, csum[g].w);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                st = 0;
                ++tn;
            }
        }
    }
}

}  // namespace ccd
