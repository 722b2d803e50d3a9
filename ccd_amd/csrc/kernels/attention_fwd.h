// attention_fwd.h - softmax(Q K^T / sqrt(d)) V for T = 256 tokens, head_dim = 64 (vision_transformer.py:85-89),
// one workgroup (4 waves) per (view, head); the whole K and V of the head live in LDS (64 KiB), so 2 WGs/CU.
//
// Everything is computed TRANSPOSED so that the softmax row of a query sits inside one lane:
//   S^T[key][q] = K . Q^T     (A = K rows from LDS, B = Q rows straight from HBM into registers)
//   a lane (q = lane & 31) then owns 128 of the 256 scores of its query, the partner lane ^ 32 the other 128:
//   the row max / row sum are in-register reductions plus ONE cross-lane exchange;
//   O^T[d][q]   = V^T . P^T   (A = V^T from LDS, B = P^T = the exponentiated accumulators, converted in place)
// The MFMA k-slot <-> key assignment of P^T is whatever the S^T accumulator layout gives
// (lanes < 32: keys {0-3, 8-11} of each 16-key group, lanes >= 32: keys {4-7, 12-15}); V^T is written to LDS
// with the same permutation, so no cross-lane data movement is needed between the two products.
// The [256,256] attention matrix is never materialised (it is only consumed by get_last_selfattention,
// vision_transformer.py:92/253-261, which is off the pretraining path).
// Saves LSE[view, head, q] = max*scale + log(sum) for the backward pass.
#pragma once

namespace ccd {

constexpr int ATT_T = 256, ATT_D = 64;
constexpr int ATT_SMEM_BYTES = 2 * ATT_T * ATT_D * 2;   // K image + V^T image = 64 KiB

// position of a 4-key chunk inside the permuted 16-key group: chunks {0,1,2,3} -> {0,2,1,3}
__device__ __forceinline__ int att_chunk_pos(int chunk) { return ((chunk & 1) << 1) | ((chunk >> 1) & 1); }

// K image: [256 keys][64 d], 128-B rows, 16-B slot XOR ((row >> 1) & 7)      (same image as the GEMM tiles)
__device__ __forceinline__ void att_stage_rows(const bf16_t* __restrict__ src, long row_stride, char* img) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = t + 256 * i, row = id >> 3, slot = id & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + (long)row * row_stride + slot * 8);
        *reinterpret_cast<u32x4*>(img + row * 128 + ((slot ^ ((row >> 1) & 7)) * 16)) = v;
    }
}
// transposed image: [64 d][256 keys] with keys permuted inside 16-groups, 512-B rows, slot XOR (d & 15)
// a thread's source rows: keys 4 kb .. 4 kb + 3 (kb = 16 w + (lane & 15)), d blocks (lane >> 4) and (lane >> 4) + 4
__device__ __forceinline__ const bf16_t* att_transposed_src(const bf16_t* __restrict__ src, long row_stride, int i, int kq) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return src + (long)(4 * (16 * w + (lane & 15)) + kq) * row_stride + ((lane >> 4) + 4 * i) * 8;
}
__device__ __forceinline__ void att_write_transposed(const u32x4 (&r)[2][4], char* img) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kb = 16 * w + (lane & 15);                  // 4-key block, keys 4*kb .. 4*kb+3
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int db = (lane >> 4) + 4 * i;               // 8-wide d block
        const int chunk = 4 * (kb >> 2) + att_chunk_pos(kb & 3);   // 8-byte chunk index inside the 512-B row
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 8 * db + j;
            const unsigned w0 = r[i][0][j >> 1], w1 = r[i][1][j >> 1], w2 = r[i][2][j >> 1], w3 = r[i][3][j >> 1];
            u32x2 o;
            o.x = perm_b32(w1, w0, (j & 1) ? 0x07060302u : 0x05040100u);
            o.y = perm_b32(w3, w2, (j & 1) ? 0x07060302u : 0x05040100u);
            const int slot = (chunk >> 1) ^ (d & 15);
            *reinterpret_cast<u32x2*>(img + d * 512 + slot * 16 + (chunk & 1) * 8) = o;
        }
    }
}
__device__ __forceinline__ void att_stage_transposed(const bf16_t* __restrict__ src, long row_stride, char* img) {
    u32x4 r[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) r[i][kq] = *reinterpret_cast<const u32x4*>(att_transposed_src(src, row_stride, i, kq));
    att_write_transposed(r, img);
}

// A lane's 16 + 16 values of its output row (columns 8 g + 4 hf .. + 3 of each 32-column tile of the transposed product) as
// 16-byte stores: the two half-waves exchange quads (v_permlane32_swap) so that lanes < 32 own columns 8 g .. + 7 for even g and
// lanes >= 32 for odd g - half the store instructions and cache-line requests of the 8-byte form (a wave store touches 32 rows
// either way).  Measured on the backward kernels: 0.36 -> 0.345 ms per layer pair.
// `live`: rows beyond the matrix take part in the exchange (both lanes of a row share the predicate) but store nothing.
__device__ __forceinline__ void att_store_row16(bf16_t* row_ptr, const f32x16 (&acc)[2], int hf, float mul, bool live = true) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            const int ge = 8 * gp, go = 8 * gp + 4;          // accumulator registers of the even / odd 8-column group
            unsigned ex = pack_bf2(acc[dt][ge + 0] * mul, acc[dt][ge + 1] * mul), ey = pack_bf2(acc[dt][ge + 2] * mul, acc[dt][ge + 3] * mul);
            unsigned ox = pack_bf2(acc[dt][go + 0] * mul, acc[dt][go + 1] * mul), oy = pack_bf2(acc[dt][go + 2] * mul, acc[dt][go + 3] * mul);
            lane32_swap(ex, ox);
            lane32_swap(ey, oy);
            // lanes < 32: (ex, ey) own quad, (ox, oy) the partner's quad of the SAME even group; lanes >= 32: (ex, ey) the
            // partner's quad of the odd group, (ox, oy) own
            const u32x4 v = {ex, ey, ox, oy};
            if (live) *reinterpret_cast<u32x4*>(row_ptr + 32 * dt + 8 * (2 * gp + hf)) = v;
        }
}

constexpr int ATT_DEPTH = 6;             // fragment reads in flight ahead of their MFMA
struct AttMapQK {          // step k = 4 kt + kk: key tile kt (32 rows x 128 B = 4 KiB apart), k-step kk
    static constexpr int reg(int k) { return k & 3; }
    static constexpr int off(int k) { return (k >> 2) * 4096; }
};
struct AttMapPV {          // step k = 4 kt + 2 s2 + dt: address register 2 kt + s2, d tile dt (32 rows x 512 B apart)
    static constexpr int reg(int k) { return k >> 1; }
    static constexpr int off(int k) { return (k & 1) * 16384; }
};

__global__ __launch_bounds__(256, 2) void attention_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int heads, float scale) {
    char* smem = dynamic_smem();
    char* k_img = smem;
    char* vt_img = smem + ATT_T * ATT_D * 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hf = lane >> 5, lq = lane & 31;
    const int view = blockIdx.x / heads, head = blockIdx.x % heads;
    const int E = heads * ATT_D;
    const long row_stride = 3L * E;
    const bf16_t* q_base = qkv + (long)view * ATT_T * row_stride + head * ATT_D;
    const bf16_t* k_base = q_base + E;
    const bf16_t* v_base = q_base + 2 * E;

    // the wave's two query tiles are requested FIRST, in front of the 16 staging loads: a workgroup is a chain of HBM round trips
    // (K / V, then q, then q again) with ~2 us of products behind each, and two workgroups per CU is all the overlap there is - loaded
    // where they were used, each query tile's round trip stood in the open
    // (hand-issued - global_load16_late: left to the compiler the eight loads sink behind the staging writes, next to their first use)
    buf_u32x4 qf2[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf2[qt][kk] = buf_u32x4{0u, 0u, 0u, 0u};
            global_load16_late(qf2[qt][kk], q_base + (long)(64 * w + 32 * qt + lq) * row_stride + 16 * kk + 8 * hf);
        }
    // ... and the V rows too: all 24 requests of a thread are out before its first wait (the K rows' writes)
    u32x4 vr[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            vr[i][kq] = u32x4{0u, 0u, 0u, 0u};
            global_load16_late(vr[i][kq], att_transposed_src(v_base, row_stride, i, kq));
        }
    att_stage_rows(k_base, row_stride, k_img);
    glds_wait_all();                       // (the hand-issued requests are older than the K rows': long there)
    vm_landed4(qf2[0]);
    vm_landed4(qf2[1]);
    vm_landed4(vr[0]);
    vm_landed4(vr[1]);
    att_write_transposed(vr, vt_img);
    __syncthreads();
    const unsigned k_addr = lds_addr_of(k_img), vt_addr = lds_addr_of(vt_img);

#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
        const int q = 64 * w + 32 * qt + lq;
        bf16x8 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, qt ? qf2[1][kk] : qf2[0][kk]);

        // Two key chunks of 128 with a running (max, sum): half the score registers of a single pass - the one-pass
        // version needed all 512 registers (and spilled), i.e. ONE workgroup per CU.
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        const float c2 = scale * 1.4426950408889634f;                  // exp(x) = 2^(x log2 e): one FMA + v_exp_f32
        float mx = -3.0e38f, sum = 0.f;
#pragma unroll 1
        for (int ch = 0; ch < 2; ++ch) {
            f32x16 s[4];
            {   // S^T chunk = K rows . Q^T: the 16 K fragments are read by hand, ATT_DEPTH ahead of their MFMA (mlp_fused.h:
                // mlp_product).  Left to the compiler every product sat behind its own read and `lgkmcnt(0)`: an LDS round trip
                // per MFMA.  Row 128 ch + 32 kt + lq: the swizzle term (row >> 1) & 7 is the lane's, tiles are immediate offsets.
                unsigned areg[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    areg[kk] = k_addr + (unsigned)(ch * 16384 + lq * 128 + ((((2 * kk + hf) ^ ((lq >> 1) & 7))) << 4));
                mlp_product<16, ATT_DEPTH, AttMapQK, MlpNoExtra>(
                    areg,
                    [&](auto K, const bf16x8& kf) {
                        constexpr int k = decltype(K)::value;
                        // (the first product of a tile takes the constant 0 as its accumulator input: 64 v_mov per chunk less)
                        if constexpr ((k & 3) == 0) s[k >> 2] = mfma_32x32x16_bf16(kf, qf[0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                        else s[k >> 2] = mfma_32x32x16_bf16(kf, qf[k & 3], s[k >> 2]);
                    },
                    [](auto) {});
            }
            float cm = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) cm = fmaxf(cm, s[kt][r]);
            cm = fmaxf(cm, shfl_xor(cm, 32));
            const float nm = fmaxf(mx, cm);
            const float alpha = fast_exp2((mx - nm) * c2);            // 0 for the first chunk
            mx = nm;
            const float mc2 = nm * c2;
            float cs4[4] = {0.f, 0.f, 0.f, 0.f};             // (four partial sums: independent chains, pairable adds)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(fmaf(s[kt][r], c2, -mc2));
                    s[kt][r] = p;
                    cs4[r & 3] += p;
                }
            float csum = (cs4[0] + cs4[1]) + (cs4[2] + cs4[3]);
            csum += shfl_xor(csum, 32);
            sum = sum * alpha + csum;
            if (ch > 0) {                                    // (the first chunk's accumulators are still zero: nothing to rescale)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            {   // O^T += V^T . P: fragment (kt, s2, dt) = V^T rows 32 dt + lq, keys 16 ks .. + 15 with ks = 2 (4 ch + kt) + s2; the
                // swizzle XORs the low bits of ks with the lane's (d & 15): one address register per ks & 7 = 2 kt + s2
                unsigned vreg[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    vreg[j] = vt_addr + (unsigned)(lq * 512 + ch * 256 + ((hf ^ (lq & 1)) << 4) + ((j ^ ((lq >> 1) & 7)) << 5));
                bf16x8 pf;
                mlp_product<16, ATT_DEPTH, AttMapPV, MlpNoExtra>(
                    vreg,
                    [&](auto K, const bf16x8& vf) {
                        constexpr int k = decltype(K)::value, kt = k >> 2, s2 = (k >> 1) & 1, dt = k & 1;
                        if constexpr (dt == 0) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) pf[e] = (short)f2bf(s[kt][8 * s2 + e]);
                        }
                        o[dt] = mfma_32x32x16_bf16(vf, pf, o[dt]);
                    },
                    [](auto) {});
            }
        }
        const float inv = 1.0f / sum;
        bf16_t* orow = out + ((long)view * ATT_T + q) * E + head * ATT_D;
        att_store_row16(orow, o, hf, inv);
        if (hf == 0) lse[((long)view * heads + head) * ATT_T + q] = mx * scale + logf(sum);
    }
}

}  // namespace ccd
