// embed.h - patch embedding (4x4/stride-4 conv == [R,48]x[48,E] GEMM) fused with bias and positional add,
// its backward, the tiny fp32 matmul used for the bicubic positional resampling (a fixed 256x256 linear map),
// bf16 column sums (bias gradients) and the fp32 -> bf16 (+ transposed) weight mirror kernel.
//   PatchEmbed.forward + prepare_tokens      Dino/modules/vision_transformer.py:128-131, 225-236
// K = 48 makes this HBM-bound: per view 48 KiB of pixels in, 256*E*4 B of fp32 tokens out.
#pragma once

namespace ccd {

constexpr int PE_PATCH = 4, PE_K = 48, PE_GW = 32, PE_GH = 8;   // 32x128 image -> 8 x 32 tokens

// one workgroup = one row of 32 patches of one view; thread e keeps W[e, 0:48] in registers
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

// d_w[e,k] += sum_tok g[tok,e]*patch[tok,k]; d_bias[e] += sum g; d_pos[tok,e] += sum_views g   (fp32 atomics)
// one workgroup = one patch row (py) of `views_per_block` consecutive views
__global__ __launch_bounds__(128) void patch_embed_bwd_kernel(const float* __restrict__ img, const float* __restrict__ g,
                                                              float* __restrict__ d_w, float* __restrict__ d_bias,
                                                              float* __restrict__ d_pos, int views,
                                                              int views_per_block, int E) {
    __shared__ float px[3][PE_PATCH][PE_GW * PE_PATCH];
    const int py = blockIdx.x % PE_GH;
    const int v0 = (blockIdx.x / PE_GH) * views_per_block;
    const int v1 = v0 + views_per_block < views ? v0 + views_per_block : views;
    for (int e0 = 0; e0 < E; e0 += blockDim.x) {
        const int e = e0 + threadIdx.x;
        float dw[PE_K];
        float dp[PE_GW];
        float db = 0.f;
#pragma unroll
        for (int k = 0; k < PE_K; ++k) dw[k] = 0.f;
#pragma unroll
        for (int k = 0; k < PE_GW; ++k) dp[k] = 0.f;
        for (int view = v0; view < v1; ++view) {
            __syncthreads();
            const float* src = img + (long)view * 3 * 32 * 128;
            for (int i = threadIdx.x; i < 3 * 4 * 128; i += blockDim.x) {
                const int c = i / 512, r = (i / 128) % 4, x = i % 128;
                px[c][r][x] = src[(long)c * 4096 + (py * 4 + r) * 128 + x];
            }
            __syncthreads();
            if (e < E) {
#pragma unroll
                for (int tx = 0; tx < PE_GW; ++tx) {
                    const float gv = g[((long)view * 256 + py * PE_GW + tx) * E + e];
                    db += gv;
                    dp[tx] += gv;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int x = 0; x < 4; ++x) dw[c * 16 + r * 4 + x] += gv * px[c][r][tx * 4 + x];
                }
            }
        }
        if (e < E) {
#pragma unroll
            for (int k = 0; k < PE_K; ++k) atomicAdd(d_w + (long)e * PE_K + k, dw[k]);
            atomicAdd(d_bias + e, db);
#pragma unroll
            for (int tx = 0; tx < PE_GW; ++tx) atomicAdd(d_pos + (long)(py * PE_GW + tx) * E + e, dp[tx]);
        }
    }
}

// C[M,N] (+)= A . B with A [M,K] (or A^T when trans_a: A stored [K,M]), B [K,N], all fp32; tiny problems only
__global__ void small_matmul_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c,
                                        int M, int N, int K, int trans_a, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += (trans_a ? a[(long)k * M + m] : a[(long)m * K + k]) * b[(long)k * N + n];
    c[(long)m * N + n] = accumulate ? c[(long)m * N + n] + acc : acc;
}

// out[n] += sum_rows x[row, n]   (bias gradients).  Block = cgn column groups (8 cols, 16 B; cgn = power of two
// <= 32 chosen by the host so narrow matrices keep all lanes busy) x 256/cgn row lanes, 4 loads in flight per thread;
// the host sizes the grid so every block streams >= 256 KiB (few same-address atomics)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long ld, int rows, int N,
                                                          const int* __restrict__ d_rows, int rows_mul,
                                                          float* __restrict__ out, int rows_per_block, int cgn_log2) {
    __shared__ float red[256][8];
    const int cgn = 1 << cgn_log2, rln = 256 >> cgn_log2;
    const int cg = threadIdx.x & (cgn - 1), rl = threadIdx.x >> cgn_log2;
    const int col = (blockIdx.x * cgn + cg) * 8;
    if (d_rows) rows = d_rows[0] * rows_mul < rows ? d_rows[0] * rows_mul : rows;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        int r = r0 + rl;
        for (; r + 3 * rln < r1; r += 4 * rln) {
            u32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const u32x4*>(x + (long)(r + u * rln) * ld + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
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
    if (rl == 0 && col < N) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float s = 0.f;
            for (int j = 0; j < rln; ++j) s += red[j * cgn + cg][k];
            atomicAdd(out + col + k, s);
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
__global__ __launch_bounds__(256) void mirror_bf16_kernel(const MirrorDesc* __restrict__ descs, int ndesc) {
    __shared__ float tile[32][33];
    int d = 0;
    while (d + 1 < ndesc && descs[d + 1].tile_begin <= (int)blockIdx.x) ++d;
    const MirrorDesc m = descs[d];
    const int tiles_c = (m.cols + 31) / 32;
    const int tl = blockIdx.x - m.tile_begin, tr = tl / tiles_c, tc = tl % tiles_c;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int i = ly; i < 32; i += 8) {
        const int r = tr * 32 + i, c = tc * 32 + lx;
        float v = 0.f;
        if (r < m.rows && c < m.cols) {
            v = m.src[(long)r * m.cols + c];
            if (m.dst) m.dst[(long)r * m.cols + c] = f2bf(v);
        }
        tile[i][lx] = v;
    }
    __syncthreads();
    if (m.dst_t) {
        for (int i = ly; i < 32; i += 8) {
            const int c = tc * 32 + i, r = tr * 32 + lx;
            if (r < m.rows && c < m.cols) m.dst_t[(long)c * m.rows + r] = f2bf(tile[lx][i]);
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
