// loss.h - DINOLoss on the device (Dino/loss/Dino_loss.py:59-143):
//   dino_loss_fwd_kernel   the two cross-view CE terms of :81-102 in ONE pass over a student row and its partner
//                          teacher row (online max/sum: probabilities are never materialised)
//   dino_loss_bwd_kernel   d/d student logits = (softmax(s/0.1) - softmax((t-c)/temp)) / 0.1 / (2M)
//   colsum_f32 + center_ema   update_center :133-143 (the all_reduce between them is the host's job)
//   seg_loss_kernel        softmax THEN cross_entropy (double softmax, :63-66 with SegLoss.cross_entropy 15-26)
// Row count M (selected character rows per view) is read from device memory.  HBM-bound: algorithmic bytes of the
// forward = 2 rows * K * 4 B per student row, of the backward the same + K * 2 B written.
#pragma once

namespace ccd {

struct OnlineLSE {
    float m, l;
    __device__ __forceinline__ void init() { m = -3.0e38f; l = 0.f; }
    __device__ __forceinline__ void add(float x) {
        if (x > m) { l = l * fast_exp(m - x) + 1.0f; m = x; }
        else l += fast_exp(x - m);
    }
    __device__ __forceinline__ void merge(float om, float ol) {
        const float nm = fmaxf(m, om);
        l = l * fast_exp(m - nm) + ol * fast_exp(om - nm);
        m = nm;
    }
};

// stats[i] = {ms, ls, mt, lt}: student row i (scaled by 1/student_temp), teacher row partner(i) (centred, / temp)
__global__ __launch_bounds__(256) void dino_loss_fwd_kernel(const float* __restrict__ s_logits,
                                                            const float* __restrict__ t_logits,
                                                            const float* __restrict__ center, int K,
                                                            const int* __restrict__ d_m, int max_rows, float inv_ts,
                                                            float inv_tt, float* __restrict__ stats,
                                                            float* __restrict__ loss_out) {
    __shared__ float red[4][5];
    const int M = d_m[0];
    const int i = blockIdx.x;
    if (i >= 2 * M || i >= max_rows) return;
    const int j = i < M ? i + M : i - M;
    const float* s = s_logits + (long)i * K;
    const float* tr = t_logits + (long)j * K;
    OnlineLSE ss, tt;
    ss.init();
    tt.init();
    float acc = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        const f32x4v sv = *reinterpret_cast<const f32x4v*>(s + k);
        const f32x4v tv = *reinterpret_cast<const f32x4v*>(tr + k);
        const f32x4v cv = *reinterpret_cast<const f32x4v*>(center + k);
        const float sx[4] = {sv.x * inv_ts, sv.y * inv_ts, sv.z * inv_ts, sv.w * inv_ts};
        const float tx[4] = {(tv.x - cv.x) * inv_tt, (tv.y - cv.y) * inv_tt, (tv.z - cv.z) * inv_tt,
                             (tv.w - cv.w) * inv_tt};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ss.add(sx[e]);
            if (tx[e] > tt.m) {
                const float f = fast_exp(tt.m - tx[e]);
                tt.l = tt.l * f + 1.0f;
                acc = acc * f + sx[e];
                tt.m = tx[e];
            } else {
                const float p = fast_exp(tx[e] - tt.m);
                tt.l += p;
                acc += p * sx[e];
            }
        }
    }
    // wave reduction of the three running quantities
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
        const float om = shfl_xor(ss.m, msk), ol = shfl_xor(ss.l, msk);
        ss.merge(om, ol);
        const float tm = shfl_xor(tt.m, msk), tl = shfl_xor(tt.l, msk), ta = shfl_xor(acc, msk);
        const float nm = fmaxf(tt.m, tm);
        const float f0 = fast_exp(tt.m - nm), f1 = fast_exp(tm - nm);
        tt.l = tt.l * f0 + tl * f1;
        acc = acc * f0 + ta * f1;
        tt.m = nm;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[w][0] = ss.m; red[w][1] = ss.l; red[w][2] = tt.m; red[w][3] = tt.l; red[w][4] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        OnlineLSE a;
        a.m = red[0][0]; a.l = red[0][1];
        float bm = red[0][2], bl = red[0][3], ba = red[0][4];
        for (int q = 1; q < 4; ++q) {
            a.merge(red[q][0], red[q][1]);
            const float nm = fmaxf(bm, red[q][2]);
            const float f0 = fast_exp(bm - nm), f1 = fast_exp(red[q][2] - nm);
            bl = bl * f0 + red[q][3] * f1;
            ba = ba * f0 + red[q][4] * f1;
            bm = nm;
        }
        stats[(long)i * 4 + 0] = a.m; stats[(long)i * 4 + 1] = a.l;
        stats[(long)i * 4 + 2] = bm;  stats[(long)i * 4 + 3] = bl;
        const float row_loss = (a.m + logf(a.l)) - ba / bl;      // -sum_k p_t * log_softmax(s)
        atomicAdd(loss_out, row_loss / (float)(2 * M));
    }
}

__global__ __launch_bounds__(256) void dino_loss_bwd_kernel(const float* __restrict__ s_logits,
                                                            const float* __restrict__ t_logits,
                                                            const float* __restrict__ center, int K,
                                                            const int* __restrict__ d_m, int max_rows, float inv_ts,
                                                            float inv_tt, const float* __restrict__ stats,
                                                            float grad_scale, const float* __restrict__ d_grad_scale,
                                                            bf16_t* __restrict__ d_logits) {
    const int M = d_m[0];
    const int i = blockIdx.x;
    if (i >= 2 * M || i >= max_rows) return;
    const int j = i < M ? i + M : i - M;
    const float ms = stats[(long)i * 4 + 0], inv_ls = 1.0f / stats[(long)i * 4 + 1];
    const float mt = stats[(long)i * 4 + 2], inv_lt = 1.0f / stats[(long)i * 4 + 3];
    const float gs = grad_scale * (d_grad_scale ? d_grad_scale[0] : 1.0f) * inv_ts / (float)(2 * M);
    const float* s = s_logits + (long)i * K;
    const float* tr = t_logits + (long)j * K;
    bf16_t* d = d_logits + (long)i * K;
    auto one = [&](const f32x4v& sv, const f32x4v& tv, const f32x4v& cv, int k) {
        float o[4];
        const float sx[4] = {sv.x, sv.y, sv.z, sv.w}, tx[4] = {tv.x - cv.x, tv.y - cv.y, tv.z - cv.z, tv.w - cv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = gs * (fast_exp(sx[e] * inv_ts - ms) * inv_ls - fast_exp(tx[e] * inv_tt - mt) * inv_lt);
        u32x2 pk;
        pk.x = pack_bf2(o[0], o[1]);
        pk.y = pack_bf2(o[2], o[3]);
        *reinterpret_cast<u32x2*>(d + k) = pk;
    };
    // two trips at a time, their six loads in front of the first store (one trip per iteration put every trip's loads behind
    // the previous trip's store: `vmcnt(0)`, the store's round trip, 64 times per row)
    int k = threadIdx.x * 4;
    for (; k + 1024 < K; k += 2048) {
        f32x4v sv[2], tv[2], cv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            sv[u] = *reinterpret_cast<const f32x4v*>(s + k + 1024 * u);
            tv[u] = *reinterpret_cast<const f32x4v*>(tr + k + 1024 * u);
            cv[u] = *reinterpret_cast<const f32x4v*>(center + k + 1024 * u);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) { needed_here(sv[u]); needed_here(tv[u]); needed_here(cv[u]); }
#pragma unroll
        for (int u = 0; u < 2; ++u) one(sv[u], tv[u], cv[u], k + 1024 * u);
    }
    for (; k < K; k += 1024)
        one(*reinterpret_cast<const f32x4v*>(s + k), *reinterpret_cast<const f32x4v*>(tr + k), *reinterpret_cast<const f32x4v*>(center + k), k);
}

// out[c] += sum over rows [r0, r1) of x[r, c]   (fp32; teacher logits -> batch centre), rows = rows_mul * d_rows[0]
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, int K,
                                                         const int* __restrict__ d_rows, int rows_mul, int max_rows,
                                                         int rows_per_block, float* __restrict__ out) {
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
    int rows = d_rows ? d_rows[0] * rows_mul : max_rows;
    rows = rows < max_rows ? rows : max_rows;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    if (k >= K || r0 >= r1) return;
    f32x4v a = {0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 3 < r1; r += 4) {                             // four rows in flight (one load per trip left the walk latency-bound)
        f32x4v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4v*>(x + (long)(r + u) * K + k);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; r < r1; ++r) {
        const f32x4v v = *reinterpret_cast<const f32x4v*>(x + (long)r * K + k);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    atomicAdd(out + k, a.x); atomicAdd(out + k + 1, a.y); atomicAdd(out + k + 2, a.z); atomicAdd(out + k + 3, a.w);
}
// out[k] += sum_d w[k, d] * v[d]   (w [K, D] bf16 row-major, D % 256 == 0, v fp32): the batch centre WITHOUT a pass over the teacher
// logits - their column sums are a matrix-vector product, sum_r (zn[r, :] . w[k, :]) = (sum_r zn[r, :]) . w[k, :] (Dino_loss.py:136:
// torch.sum(teacher_output, dim=0) of teacher_output = zn @ w^T).  Half a wave per row: 32 lanes x 8 elements per 256-wide step.
__global__ __launch_bounds__(256) void matvec_bf16_kernel(const bf16_t* __restrict__ w, long ldw, const float* __restrict__ v, int K, int D,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63, hf = lane >> 5, lq = lane & 31;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int k = 2 * wave + hf; k < K; k += 2 * nwaves) {
        float acc = 0.f;
        for (int d0 = 0; d0 < D; d0 += 256) {
            const u32x4 pk = *reinterpret_cast<const u32x4*>(w + (long)k * ldw + d0 + 8 * lq);
            const f32x4v v0 = *reinterpret_cast<const f32x4v*>(v + d0 + 8 * lq), v1 = *reinterpret_cast<const f32x4v*>(v + d0 + 8 * lq + 4);
            float x[8];
            unpack8(pk, x);
            acc += (x[0] * v0.x + x[1] * v0.y) + (x[2] * v0.z + x[3] * v0.w) + (x[4] * v1.x + x[5] * v1.y) + (x[6] * v1.z + x[7] * v1.w);
        }
#pragma unroll
        for (int msk = 16; msk >= 1; msk >>= 1) acc += shfl_xor(acc, msk);
        if (lq == 0) out[k] += acc;
    }
}
// center = center*momentum + (batch_sum / (2M * world)) * (1 - momentum)        Dino_loss.py:140-143
__global__ void center_ema_kernel(float* __restrict__ center, const float* __restrict__ batch_sum, int K,
                                  const int* __restrict__ d_m, int world, float momentum) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float denom = (float)(2 * d_m[0]) * (float)world;
    center[k] = center[k] * momentum + (batch_sum[k] / denom) * (1.0f - momentum);
}

// seg loss: p = softmax(logits over the 2 classes); loss = mean CE(p, target)  (CE applies log_softmax AGAIN)
// logits [images, 2, 4096] fp32; target = mask_a (float, first `half` images) / idmap_b (uint8 0=text, 255=bg)
constexpr int SEG_LOSS_PER_THREAD = 8;
__global__ __launch_bounds__(256) void seg_loss_kernel(const float* __restrict__ logits, const float* __restrict__ mask_a,
                                                       const unsigned char* __restrict__ idmap_b, int half,
                                                       float grad_scale, float* __restrict__ loss_out,
                                                       float* __restrict__ d_logits) {
    __shared__ float red[4];
    const long npix = 2L * half * CM_PIX;
    float loss = 0.f;
    // SEG_LOSS_PER_THREAD pixels per thread: the loss is ONE address, and a block's atomic add to it is the kernel's critical path -
    // 8 192 blocks of one pixel per thread took 108 us for 16 MB of logits
#pragma unroll
    for (int u = 0; u < SEG_LOSS_PER_THREAD; ++u) {
        const long i = ((long)blockIdx.x * SEG_LOSS_PER_THREAD + u) * 256 + threadIdx.x;
        if (i >= npix) break;
        const long img = i / CM_PIX, pix = i % CM_PIX;
        const float l0 = logits[(img * 2) * CM_PIX + pix], l1 = logits[(img * 2 + 1) * CM_PIX + pix];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const int tgt = img < half ? (mask_a[img * CM_PIX + pix] != 0.0f ? 1 : 0)
                                   : (idmap_b[(img - half) * CM_PIX + pix] != CM_BG ? 1 : 0);
        const float pm = fmaxf(p0, p1);
        const float lse = pm + logf(expf(p0 - pm) + expf(p1 - pm));
        loss += lse - (tgt ? p1 : p0);
        if (d_logits) {
            const float q1 = expf(p1 - lse);                         // second softmax, class 1
            const float dd = (q1 - (float)tgt) * 2.0f * p0 * p1 * grad_scale / (float)npix;
            d_logits[(img * 2 + 1) * CM_PIX + pix] = dd;
            d_logits[(img * 2) * CM_PIX + pix] = -dd;
        }
    }
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_out, (red[0] + red[1] + red[2] + red[3]) / (float)npix);
}

}  // namespace ccd
