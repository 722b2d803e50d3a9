// gemm_nt32.h - the occupancy-oriented NT GEMM: 128x128 output tile, BK = 32, two 16-KiB LDS stages (32 KiB per
// workgroup), <= 128 VGPRs, accumulators produced transposed and stored straight from registers.
//
// Why a second NT kernel: on the transformer's shapes (K = 384 ... 1536, huge M) the 64-KiB / 256-VGPR tile of gemm.h
// leaves only 2 workgroups per CU, all waves of a workgroup march in lockstep between barriers, and an ablation of the
// A-resident variant (gemm_ares.h) showed its phases do not overlap at all (stores + loads + LDS writes + barrier +
// MFMA simply add up).  Here 4-5 independent workgroups share a CU, so while one sits in a barrier or a store burst
// the others keep the matrix pipe busy.  Same swizzle idea as gemm.h on 64-byte rows: slot ^ ((row >> 2) & 3)
// (brute-force checked: ds_read_b128 fragment reads and ds_write_b128 staging writes conflict-free).
#pragma once

namespace ccd {

constexpr int NT32_BK = 32;
constexpr int NT32_STAGE_BYTES = 2 * 128 * NT32_BK * 2;          // A + B chunk of one stage: 16 KiB
constexpr int NT32_SMEM_BYTES = 2 * NT32_STAGE_BYTES;            // 32 KiB

__device__ __forceinline__ int nt32_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) * 16); }

template <int EPI>
__device__ __forceinline__ f32x4v nt32_store4(const GemmParams& p, int gm, int gn, float v0, float v1, float v2, float v3,
                                              float scale) {
    if (EPI != EPI_DGELU && EPI != EPI_ATOMIC && p.bias) {
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
    } else if (EPI == EPI_F32) {
        const f32x4v o = {v0, v1, v2, v3};
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
__global__ __launch_bounds__(256, 4) void gemm_nt32_kernel(GemmParams p) {
    const int m_static = p.M;
    if (p.d_rows) {
        const int dyn = p.d_rows[0] * p.rows_mul;
        p.M = dyn < p.M ? dyn : p.M;
    }
    char* smem = dynamic_smem();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, hf = lane >> 5, lq = lane & 31;
    const int wm = w >> 1, wn = w & 1;
    const int tiles_m = (m_static + 127) / 128, tiles_n = (p.N + 127) / 128;
    const unsigned tile = xcd_remap(blockIdx.x, (unsigned)(tiles_m * tiles_n));
    int tm, tn;
    if (p.m_fastest) { tm = tile % tiles_m; tn = tile / tiles_m; }
    else { tn = tile % tiles_n; tm = tile / tiles_n; }
    const int m0 = tm * 128, n0 = tn * 128;
    if (m0 >= p.M) return;

    // staging: thread t moves rows (t >> 2) and (t >> 2) + 64 of each operand, 16-byte slot t & 3; rows beyond the
    // matrix are clamped to the last valid one (their products end up in rows / columns that are never stored)
    const int srow = t >> 2, sslot = t & 3;
    const bf16_t* pa[2];
    const bf16_t* pb[2];
    int wr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int ra = m0 + srow + 64 * h, rb = n0 + srow + 64 * h;
        ra = ra < p.M ? ra : p.M - 1;
        rb = rb < p.N ? rb : p.N - 1;
        pa[h] = p.A + (long)ra * p.lda + sslot * 8;
        pb[h] = p.B + (long)rb * p.ldb + sslot * 8;
        wr[h] = nt32_off(srow + 64 * h, sslot);
    }
    int a_off[2][2], b_off[2][2];                           // [tile][kk] fragment offsets inside a chunk
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            a_off[i][kk] = nt32_off(64 * wm + 32 * i + lq, 2 * kk + hf);
            b_off[i][kk] = nt32_off(64 * wn + 32 * i + lq, 2 * kk + hf) + 8192;
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / NT32_BK;
    u32x4 ra[2], rb[2];
    auto fetch = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ra[h] = *reinterpret_cast<const u32x4*>(pa[h]);
            rb[h] = *reinterpret_cast<const u32x4*>(pb[h]);
            pa[h] += NT32_BK;
            pb[h] += NT32_BK;
        }
    };
    auto commit = [&](int stage) {
        char* dst = smem + stage * NT32_STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<u32x4*>(dst + wr[h]) = ra[h];
            *reinterpret_cast<u32x4*>(dst + 8192 + wr[h]) = rb[h];
        }
    };
    fetch();
    commit(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char* src = smem + (kt & 1) * NT32_STAGE_BYTES;
        if (kt + 1 < nk) fetch();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(src + a_off[i][kk]);
                b[i] = *reinterpret_cast<const bf16x8*>(src + b_off[i][kk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16_bf16(b[j], a[i], acc[i][j]);   // D[n][m]
        }
        if (kt + 1 < nk) commit((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue straight from registers: lane owns row (lq) of tile i and 4 consecutive columns per register group
    f32x4v csum[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) { csum[j][g].x = 0.f; csum[j][g].y = 0.f; csum[j][g].z = 0.f; csum[j][g].w = 0.f; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gm = m0 + 64 * wm + 32 * i + lq;
        if (gm < p.M) {
            float scale = 1.0f;
            if (EPI == EPI_RESID && p.rowscale) scale = p.rowscale[gm / p.rows_per_sample];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int gn = n0 + 64 * wn + 32 * j + 8 * g + 4 * hf;
                    if (gn < p.N)
                        csum[j][g] += nt32_store4<EPI>(p, gm, gn, acc[i][j][4 * g] * p.alpha, acc[i][j][4 * g + 1] * p.alpha,
                                                       acc[i][j][4 * g + 2] * p.alpha, acc[i][j][4 * g + 3] * p.alpha, scale);
                }
        }
    }
    if ((EPI == EPI_DGELU || EPI == EPI_BF16) && p.colsum) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int msk = 16; msk >= 1; msk >>= 1) {
                    csum[j][g].x += shfl_xor(csum[j][g].x, msk); csum[j][g].y += shfl_xor(csum[j][g].y, msk);
                    csum[j][g].z += shfl_xor(csum[j][g].z, msk); csum[j][g].w += shfl_xor(csum[j][g].w, msk);
                }
                const int gn = n0 + 64 * wn + 32 * j + 8 * g + 4 * hf;
                if (lq == 0 && gn < p.N) {
                    atomicAdd(p.colsum + gn, csum[j][g].x); atomicAdd(p.colsum + gn + 1, csum[j][g].y);
                    atomicAdd(p.colsum + gn + 2, csum[j][g].z); atomicAdd(p.colsum + gn + 3, csum[j][g].w);
                }
            }
    }
}

}  // namespace ccd
