// embed.h - patch embedding (4x4/stride-4 conv == [R,48]x[48,E] GEMM) fused with bias and positional add,
// its backward, the tiny fp32 matmul used for the bicubic positional resampling (a fixed 256x256 linear map),
// bf16 column sums (bias gradients) and the fp32 -> bf16 (+ transposed) weight mirror kernel.
//   PatchEmbed.forward + prepare_tokens      Dino/modules/vision_transformer.py:128-131, 225-236
// K = 48 makes this HBM-bound: per view 48 KiB of pixels in, 256*E*4 B of fp32 tokens out.
#pragma once

namespace ccd {

constexpr int PE_PATCH = 4, PE_K = 48, PE_GW = 32, PE_GH = 8;   // 32x128 image -> 8 x 32 tokens

// one workgroup = one row of 32 patches of one view; thread e keeps W[e, 0:48] in registers
// (round 6, measured and not kept - profiles/r06_patch_embed_lab.jsonl: the pixels by scalar loads, s_load_dwordx16 per four tokens and no
// LDS: 136 us against 113; three columns per thread, so that a token's 12 broadcast LDS reads feed 144 fused multiply-adds: 122 against 123)
__global__ __launch_bounds__(128) void patch_embed_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ pos, float* __restrict__ out,
                                                              int E) {
    __shared__ float px[3][PE_PATCH][PE_GW * PE_PATCH];
    const int view = blockIdx.x / PE_GH, py = blockIdx.x % PE_GH;
    const float* src = img + (long)view * 3 * 32 * 128;
    for (int i = threadIdx.x; i < 3 * 4 * 128; i += blockDim.x) {
        const int c = i / 512, r = (i / 128) % 4, x = i % 128;
        px[c][r][x] = src[(long)c * 4096 + (py * 4 + r) * 128 + x];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float wr[PE_K];
#pragma unroll
        for (int k = 0; k < PE_K; ++k) wr[k] = w[(long)e * PE_K + k];   // conv weight [E,3,4,4] flattened
        const float b = bias[e];
        for (int tx = 0; tx < PE_GW; ++tx) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int x = 0; x < 4; ++x) acc += wr[c * 16 + r * 4 + x] * px[c][r][tx * 4 + x];
            const int tok = py * PE_GW + tx;
            out[((long)view * 256 + tok) * E + e] = acc + b + pos[(long)tok * E + e];
        }
    }
}

// Backward of the patch embedding, as three streaming passes + one MFMA product (host: ccd_patch_embed_bwd):
//   pos_grad_cast_kernel : d_pos[tok,e] += sum_views g[view,tok,e]  and  gb = bf16(g)       (one read of g)
//   colsum_bf16_kernel   : d_bias += column sums of gb
//   patch_rows_kernel    : patches[tok, c*16 + r*4 + x] = bf16(img[view, c, 4*py + r, 4*px + x])  (im2col, K = 48)
//   gemm_bf16_kernel<TN> : d_w[E,48] += gb^T . patches
// thread = (token, 4 consecutive channels), blockIdx.y = slice of the views; 4 loads in flight
// (round 6) the bf16 gradient stream needs no copy: d_pos only
__global__ __launch_bounds__(256) void pos_grad_sum16_kernel(const bf16_t* __restrict__ g, float* __restrict__ d_pos, int views, int E,
                                                             int views_per_slice) {
    const int e4 = E >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 256 * e4) return;
    const int tok = i / e4, e = (i % e4) * 4;
    const int v0 = blockIdx.y * views_per_slice;
    const int v1 = v0 + views_per_slice < views ? v0 + views_per_slice : views;
    const long stride = 256L * E;
    const long base = (long)tok * E + e;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    int v = v0;
    for (; v + 3 < v1; v += 4) {
        u32x2 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const u32x2*>(g + base + (v + u) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += f32x4v{bf_lo(x[u].x), bf_hi(x[u].x), bf_lo(x[u].y), bf_hi(x[u].y)};
    }
    for (; v < v1; ++v) {
        const u32x2 x = *reinterpret_cast<const u32x2*>(g + base + v * stride);
        acc += f32x4v{bf_lo(x.x), bf_hi(x.x), bf_lo(x.y), bf_hi(x.y)};
    }
    atomicAdd(d_pos + base, acc.x);
    atomicAdd(d_pos + base + 1, acc.y);
    atomicAdd(d_pos + base + 2, acc.z);
    atomicAdd(d_pos + base + 3, acc.w);
}
__global__ __launch_bounds__(256) void pos_grad_cast_kernel(const float* __restrict__ g, float* __restrict__ d_pos,
                                                            bf16_t* __restrict__ gb, int views, int E,
                                                            int views_per_slice) {
    const int e4 = E >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 256 * e4) return;
    const int tok = i / e4, e = (i % e4) * 4;
    const int v0 = blockIdx.y * views_per_slice;
    const int v1 = v0 + views_per_slice < views ? v0 + views_per_slice : views;
    const long stride = 256L * E;
    const long base = (long)tok * E + e;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    int v = v0;
    for (; v + 3 < v1; v += 4) {
        f32x4v x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const f32x4v*>(g + base + (v + u) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc += x[u];
            u32x2 o;
            o.x = pack_bf2(x[u].x, x[u].y);
            o.y = pack_bf2(x[u].z, x[u].w);
            *reinterpret_cast<u32x2*>(gb + base + (v + u) * stride) = o;
        }
    }
    for (; v < v1; ++v) {
        const f32x4v x = *reinterpret_cast<const f32x4v*>(g + base + v * stride);
        acc += x;
        u32x2 o;
        o.x = pack_bf2(x.x, x.y);
        o.y = pack_bf2(x.z, x.w);
        *reinterpret_cast<u32x2*>(gb + base + v * stride) = o;
    }
    atomicAdd(d_pos + base, acc.x);
    atomicAdd(d_pos + base + 1, acc.y);
    atomicAdd(d_pos + base + 2, acc.z);
    atomicAdd(d_pos + base + 3, acc.w);
}

// thread = (token, channel c, patch row r): 4 pixels (16 B) in, 4 bf16 (8 B) out
__global__ __launch_bounds__(256) void patch_rows_kernel(const float* __restrict__ img, bf16_t* __restrict__ patches,
                                                         long tokens) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= tokens * 12) return;
    const long tok = i / 12;
    const int cr = (int)(i % 12), c = cr >> 2, r = cr & 3;
    const long view = tok >> 8;
    const int py = (int)(tok >> 5) & 7, px = (int)tok & 31;
    const f32x4v v = *reinterpret_cast<const f32x4v*>(img + ((view * 3 + c) * 32 + py * 4 + r) * 128 + px * 4);
    u32x2 o;
    o.x = pack_bf2(v.x, v.y);
    o.y = pack_bf2(v.z, v.w);
    *reinterpret_cast<u32x2*>(patches + tok * PE_K + c * 16 + r * 4) = o;
}

// C[M,N] (+)= A . B with A [M,K] (or A^T when trans_a: A stored [K,M]), B [K,N], all fp32; tiny problems only
__global__ void small_matmul_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c,
                                        int M, int N, int K, int trans_a, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;   // four independent chains (k mod 4), each in increasing k
    int k = 0;
    // 32 k-steps of loads in flight per thread: with 4 the loop waited ~0.6 us of L2 latency 64 times (39 us for the 256 x 256 x 384
    // positional resample, three launches per step); same chains, same order, same bits
    for (; k + 31 < K; k += 32) {
        float av[32], bv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) av[u] = trans_a ? a[(long)(k + u) * M + m] : a[(long)m * K + k + u];
#pragma unroll
        for (int u = 0; u < 32; ++u) bv[u] = b[(long)(k + u) * N + n];
#pragma unroll
        for (int u = 0; u < 32; u += 4) {
            acc0 += av[u] * bv[u];
            acc1 += av[u + 1] * bv[u + 1];
            acc2 += av[u + 2] * bv[u + 2];
            acc3 += av[u + 3] * bv[u + 3];
        }
    }
    for (; k + 3 < K; k += 4) {
        const float a0 = trans_a ? a[(long)k * M + m] : a[(long)m * K + k];
        const float a1 = trans_a ? a[(long)(k + 1) * M + m] : a[(long)m * K + k + 1];
        const float a2 = trans_a ? a[(long)(k + 2) * M + m] : a[(long)m * K + k + 2];
        const float a3 = trans_a ? a[(long)(k + 3) * M + m] : a[(long)m * K + k + 3];
        acc0 += a0 * b[(long)k * N + n];
        acc1 += a1 * b[(long)(k + 1) * N + n];
        acc2 += a2 * b[(long)(k + 2) * N + n];
        acc3 += a3 * b[(long)(k + 3) * N + n];
    }
    for (; k < K; ++k) acc0 += (trans_a ? a[(long)k * M + m] : a[(long)m * K + k]) * b[(long)k * N + n];
    const float acc = (acc0 + acc1) + (acc2 + acc3);
    c[(long)m * N + n] = accumulate ? c[(long)m * N + n] + acc : acc;
}

// out[n] += sum_rows x[row, n]   (bias gradients).  Block = 1024 threads = cgn column groups (8 cols, 16 B; cgn = power
// of two <= 32 chosen by the host so narrow matrices keep all lanes busy) x 1024/cgn row lanes, 8 loads in flight per
// thread.  Big blocks on purpose: the fp32 atomics that publish a block's sums cost ~0.3 ns each chip-wide (measured:
// 131 k atomics = 44 us), so the grid is sized for few blocks (2 per CU) rather than many.
constexpr int COLSUM_THREADS = 1024;
__global__ __launch_bounds__(COLSUM_THREADS) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long ld, int rows, int N,
                                                          const int* __restrict__ d_rows, int rows_mul,
                                                          float* __restrict__ out, int rows_per_block, int cgn_log2) {
    __shared__ float red[COLSUM_THREADS][8];
    const int cgn = 1 << cgn_log2, rln = COLSUM_THREADS >> cgn_log2;
    const int cg = threadIdx.x & (cgn - 1), rl = threadIdx.x >> cgn_log2;
    const int col = (blockIdx.x * cgn + cg) * 8;
    if (d_rows) rows = d_rows[0] * rows_mul < rows ? d_rows[0] * rows_mul : rows;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        int r = r0 + rl;
        for (; r + 7 * rln < r1; r += 8 * rln) {              // 8 x 16 B in flight per thread
            u32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const u32x4*>(x + (long)(r + u * rln) * ld + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float v[8];
                unpack8(w[u], v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
        }
        for (; r < r1; r += rln) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + (long)r * ld + col), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    // thread t < 8*cgn publishes column t of this block's column range: consecutive lanes -> consecutive addresses
    if ((int)threadIdx.x < 8 * cgn) {
        const int c_local = threadIdx.x, g = c_local >> 3, k = c_local & 7;
        const int c = blockIdx.x * cgn * 8 + c_local;
        if (c < N) {
            float s = 0.f;
            for (int j = 0; j < rln; ++j) s += red[j * cgn + g][k];
            atomicAdd(out + c, s);
        }
    }
}

// bf16 mirrors of a batch of fp32 matrices: dst[r,c] = bf16(src[r,c]) and (optionally) dst_t[c,r]
struct MirrorDesc {
    const float* src;
    bf16_t* dst;       // [rows, cols] or null
    bf16_t* dst_t;     // [cols, rows] or null
    int rows, cols;
    int tile_begin;    // first 32x32 tile index of this matrix in the launch
};
// 64 x 64 tiles (MIRROR_TILE; round 5: 32 x 32 tiles with 2-byte stores = 64-byte segments of the transposed copy ran at 1.6 TB/s):
// rows of 256 bytes in, 128-byte segments out on both copies - the transposed one as 4-byte pairs of rows
constexpr int MIRROR_TILE = 64;
__global__ __launch_bounds__(256) void mirror_bf16_kernel(const MirrorDesc* __restrict__ descs, int ndesc) {
    __shared__ float tile[MIRROR_TILE][MIRROR_TILE + 1];
    int d = 0;
    while (d + 1 < ndesc && descs[d + 1].tile_begin <= (int)blockIdx.x) ++d;
    const MirrorDesc m = descs[d];
    const int tiles_c = (m.cols + MIRROR_TILE - 1) / MIRROR_TILE;
    const int tl = blockIdx.x - m.tile_begin, tr = tl / tiles_c, tc = tl % tiles_c;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int i = ly; i < MIRROR_TILE; i += 4) {
        const int r = tr * MIRROR_TILE + i, c = tc * MIRROR_TILE + lx;
        float v = 0.f;
        if (r < m.rows && c < m.cols) {
            v = m.src[(long)r * m.cols + c];
            if (m.dst) m.dst[(long)r * m.cols + c] = f2bf(v);
        }
        tile[i][lx] = v;
    }
    __syncthreads();
    if (m.dst_t) {
        const int rp = threadIdx.x & 31, cy = threadIdx.x >> 5;                      // rows 2 rp, 2 rp + 1 of the tile
        const int r = tr * MIRROR_TILE + 2 * rp;
        const bool pairs = (m.rows & 1) == 0;                                        // (4-byte alignment of every transposed row)
        for (int i = cy; i < MIRROR_TILE; i += 8) {
            const int c = tc * MIRROR_TILE + i;
            if (c >= m.cols) break;
            bf16_t* o = m.dst_t + (long)c * m.rows + r;
            if (pairs && r + 1 < m.rows) *reinterpret_cast<unsigned*>(o) = pack_bf2(tile[2 * rp][i], tile[2 * rp + 1][i]);
            else {
                if (r < m.rows) o[0] = f2bf(tile[2 * rp][i]);
                if (r + 1 < m.rows) o[1] = f2bf(tile[2 * rp + 1][i]);
            }
        }
    }
}

// dst(bf16)[r, :] = src(f32)[r, :] * rowscale[r / rows_per_sample]: the residual-gradient stream entering a branch
// (the DropPath scale of that branch, vision_transformer.py:27-35, applied on the way back)
__global__ __launch_bounds__(256) void scale_cast_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                              const float* __restrict__ rowscale, int rows_per_sample,
                                                              long rows, int E) {
    const int e4 = E >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * e4) return;
    const long r = i / e4;
    const int c = (int)(i % e4) * 4;
    const float s = rowscale ? rowscale[r / rows_per_sample] : 1.0f;
    const f32x4v v = *reinterpret_cast<const f32x4v*>(src + r * E + c);
    u32x2 o;
    o.x = pack_bf2(v.x * s, v.y * s);
    o.y = pack_bf2(v.z * s, v.w * s);
    *reinterpret_cast<u32x2*>(dst + r * E + c) = o;
}

// plain fp32 -> bf16 cast of a flat range (used to mirror whole parameter arenas)
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4v v = *reinterpret_cast<const f32x4v*>(src + i);
        u32x2 o;
        o.x = pack_bf2(v.x, v.y);
        o.y = pack_bf2(v.z, v.w);
        *reinterpret_cast<u32x2*>(dst + i) = o;
    } else {
        for (long k = i; k < n; ++k) dst[k] = f2bf(src[k]);
    }
}

}  // namespace ccd
