// optim.h - the parameter update of train.py:244-272 as three multi-tensor kernels over the flat parameter arena:
//   seg_sumsq_kernel   per-tensor squared gradient norms            (clip_gradients, modules/utils.py:132-141)
//   adamw_kernel       per-TENSOR clip (coef = clip/(norm+1e-6) if < 1) fused with torch.optim.AdamW's update
//                      (train.py:133, defaults betas (0.9,0.999), eps 1e-8) and the bf16 mirror refresh
//   ema_kernel         teacher = m*teacher + (1-m)*student          (train.py:264-272) + teacher bf16 mirror
// A "segment" is one parameter tensor; the arena is cut into chunks of <= 1024 elements that never straddle a
// segment (tables built on the host once).  HBM-bound: AdamW touches 4+4+4 B read, 4+4+4+2 B written per element.
#pragma once

namespace ccd {

constexpr int OPT_CHUNK = 1024;

struct SegHyper {        // per segment, refreshed by the host every iteration
    float lr_wd;         // lr * weight_decay (0 for biases / 1-D tensors, modules/utils.py:643-654)
    float step_size;     // lr / (1 - beta1^t)
    float inv_sqrt_bc2;  // 1 / sqrt(1 - beta2^t)
    float active;        // 0: tensor has no gradient this iteration (unused param, or cancelled last layer)
};

// one workgroup walks `chunks_per_block` consecutive chunks and publishes a running sum only when the segment changes:
// one block per chunk meant 16 k same-address atomics for the 16.8 M-element last layer alone (0.29 ms for 183 MB)
__global__ __launch_bounds__(256) void seg_sumsq_kernel(const float* __restrict__ grad, const int* __restrict__ chunk_seg,
                                                        const long* __restrict__ chunk_begin,
                                                        const int* __restrict__ chunk_len, float* __restrict__ norm2,
                                                        int nchunks, int chunks_per_block) {
    __shared__ float red[4];
    const int c0 = blockIdx.x * chunks_per_block;
    const int c1 = c0 + chunks_per_block < nchunks ? c0 + chunks_per_block : nchunks;
    float s = 0.f;
    int seg = c0 < nchunks ? chunk_seg[c0] : -1;
    for (int c = c0; c < c1; ++c) {
        const int cseg = chunk_seg[c];
        if (cseg != seg) {                                   // wave-uniform: flush the finished segment's partial sum
            s = wave_sum(s);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(norm2 + seg, red[0] + red[1] + red[2] + red[3]);
            __syncthreads();
            s = 0.f;
            seg = cseg;
        }
        const long base = chunk_begin[c];                    // 64-element aligned (arena.py ALIGN)
        const int len = chunk_len[c];
        const int i4 = threadIdx.x * 4;
        if (i4 + 3 < len) {
            const f32x4v g = *reinterpret_cast<const f32x4v*>(grad + base + i4);
            s += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
        } else {
            for (int i = i4; i < len && i < i4 + 4; ++i) { const float g = grad[base + i]; s += g * g; }
        }
    }
    if (seg >= 0) {
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(norm2 + seg, red[0] + red[1] + red[2] + red[3]);
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                    float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                    bf16_t* __restrict__ mirror, const int* __restrict__ chunk_seg,
                                                    const long* __restrict__ chunk_begin,
                                                    const int* __restrict__ chunk_len,
                                                    const SegHyper* __restrict__ hyper,
                                                    const float* __restrict__ norm2, float clip, float beta1,
                                                    float beta2, float eps) {
    const int c = blockIdx.x;
    const int seg = chunk_seg[c];
    const SegHyper h = hyper[seg];
    if (h.active == 0.0f) return;
    float coef = 1.0f;
    if (clip > 0.f) {
        const float cc = clip / (sqrtf(norm2[seg]) + 1e-6f);
        if (cc < 1.0f) coef = cc;
    }
    const long base = chunk_begin[c];                        // 64-element aligned (arena.py ALIGN): 16-byte accesses
    const int len = chunk_len[c];
    auto one = [&](float g, float& p, float& m, float& v) {
        g *= coef;
        p *= 1.0f - h.lr_wd;
        m = m * beta1 + (1.0f - beta1) * g;
        v = v * beta2 + (1.0f - beta2) * g * g;
        const float denom = sqrtf(v) * h.inv_sqrt_bc2 + eps;
        p -= h.step_size * (m / denom);
    };
    for (int i4 = threadIdx.x * 4; i4 < len; i4 += 1024) {
        const long k = base + i4;
        if (i4 + 3 < len) {                                  // four 16-byte loads in flight, then three 16-byte stores (+ 8 bytes)
            const f32x4v gq = *reinterpret_cast<const f32x4v*>(grad + k);
            const f32x4v pq = *reinterpret_cast<const f32x4v*>(param + k);
            const f32x4v mq = *reinterpret_cast<const f32x4v*>(exp_avg + k);
            const f32x4v vq = *reinterpret_cast<const f32x4v*>(exp_avg_sq + k);
            const float g[4] = {gq.x, gq.y, gq.z, gq.w};
            float p[4] = {pq.x, pq.y, pq.z, pq.w}, m[4] = {mq.x, mq.y, mq.z, mq.w}, v[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) one(g[e], p[e], m[e], v[e]);
            *reinterpret_cast<f32x4v*>(param + k) = f32x4v{p[0], p[1], p[2], p[3]};
            *reinterpret_cast<f32x4v*>(exp_avg + k) = f32x4v{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<f32x4v*>(exp_avg_sq + k) = f32x4v{v[0], v[1], v[2], v[3]};
            if (mirror) {
                u32x2 o;
                o.x = pack_bf2(p[0], p[1]);
                o.y = pack_bf2(p[2], p[3]);
                *reinterpret_cast<u32x2*>(mirror + k) = o;
            }
        } else {
            for (int i = i4; i < len; ++i) {
                float p = param[base + i], m = exp_avg[base + i], v = exp_avg_sq[base + i];
                one(grad[base + i], p, m, v);
                param[base + i] = p; exp_avg[base + i] = m; exp_avg_sq[base + i] = v;
                if (mirror) mirror[base + i] = f2bf(p);
            }
        }
    }
}

// in-place per-tensor clip only (when an external optimizer is used): grad *= min(1, clip/(norm+1e-6))
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ grad, const int* __restrict__ chunk_seg,
                                                         const long* __restrict__ chunk_begin,
                                                         const int* __restrict__ chunk_len,
                                                         const float* __restrict__ norm2, float clip) {
    const int c = blockIdx.x;
    const float cc = clip / (sqrtf(norm2[chunk_seg[c]]) + 1e-6f);
    if (!(cc < 1.0f)) return;
    const long base = chunk_begin[c];
    for (int i = threadIdx.x; i < chunk_len[c]; i += 256) grad[base + i] *= cc;
}

// d_m (optional, device): {m, 1 - m} read at run time instead of the launch arguments - a HIP graph of the training step
// replays with the momentum of ITS iteration (train.py:264: the schedule changes every iteration)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ teacher, const float* __restrict__ student,
                                                  bf16_t* __restrict__ mirror, long n, float m, float om,
                                                  const float* __restrict__ d_m) {
    if (d_m) { m = d_m[0]; om = d_m[1]; }
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        f32x4v t = *reinterpret_cast<f32x4v*>(teacher + i);
        const f32x4v s = *reinterpret_cast<const f32x4v*>(student + i);
        t.x = t.x * m + om * s.x; t.y = t.y * m + om * s.y; t.z = t.z * m + om * s.z; t.w = t.w * m + om * s.w;
        *reinterpret_cast<f32x4v*>(teacher + i) = t;
        if (mirror) {
            u32x2 o;
            o.x = pack_bf2(t.x, t.y);
            o.y = pack_bf2(t.z, t.w);
            *reinterpret_cast<u32x2*>(mirror + i) = o;
        }
    } else {
        for (long k = i; k < n; ++k) {
            const float t = teacher[k] * m + om * student[k];
            teacher[k] = t;
            if (mirror) mirror[k] = f2bf(t);
        }
    }
}

}  // namespace ccd
