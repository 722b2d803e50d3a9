// decoder.h - the non-GEMM kernels of the FINETUNE path: DINO_Finetune = ViT encoder + Mlp +
// NRTR transformer decoder + TFLoss (reference: Dino/model/dino_vision.py:134-246, Dino/decoder/nrtr_decoder.py:92-170,
// Dino/decoder/transformer_module.py:8-97, Dino/decoder/transformer_layers.py:150-163, Dino/loss/ce_loss.py:94-128).
//
//   dropout_kernel          nn.Dropout as a counter-based mask (no mask tensor: backward regenerates it from the seed)
//   dec_embed_fwd/bwd       trg_word_emb(seq) + position_table[:T]  (nrtr_decoder.py:93-95), padding_idx row frozen
//   dec_attn_fwd/bwd        MultiHeadAttention core for SHORT query sequences (T <= 32 queries, <= 256 keys, d_k = 64):
//                           self-attention with the pad & causal mask and encoder-decoder attention over the 256 tokens
//   tf_loss_fwd/bwd         TFLoss: cross entropy of logits[:, :-1] vs targets[:, 1:], <PAD> ignored, mean
//   greedy_step             softmax + argmax of one decoding position (nrtr_decoder.py:160-168)
// All of these are small next to the encoder (25 target positions against 256 image tokens per sample); they are plain
// fp32 SIMT kernels, HBM-bound: the decoder attention reads each K/V row of a (sample, head) exactly once.
#pragma once

namespace ccd {

// ------------------------------------------------------------------------------------------------ dropout
// keep(idx) is a pure function of (seed, element index): splitmix64 finaliser, top 32 bits compared with p * 2^32.
__device__ __forceinline__ bool drop_keep(unsigned long long seed, unsigned long long idx, unsigned thr) {
    unsigned long long z = idx + seed;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (unsigned)(z >> 32) >= thr;
}

// dst = (resid ? resid : 0) + keep(i) * scale * src     (4 elements per thread; n % 4 == 0)
template <bool SRC_BF16, bool DST_BF16>
__global__ __launch_bounds__(256) void dropout_kernel(const void* __restrict__ src_, const float* __restrict__ resid,
                                                      void* __restrict__ dst_, long n, unsigned long long seed,
                                                      unsigned thr, float scale) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float v[4];
    if (SRC_BF16) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(src_) + i);
        v[0] = bf_lo(w.x); v[1] = bf_hi(w.x); v[2] = bf_lo(w.y); v[3] = bf_hi(w.y);
    } else {
        const f32x4v w = *reinterpret_cast<const f32x4v*>(reinterpret_cast<const float*>(src_) + i);
        v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (thr == 0u || drop_keep(seed, (unsigned long long)(i + e), thr)) ? v[e] * scale : 0.f;
    if (resid) {
        const f32x4v r = *reinterpret_cast<const f32x4v*>(resid + i);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    if (DST_BF16) {
        u32x2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(dst_) + i) = o;
    } else {
        f32x4v o;
        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
        *reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(dst_) + i) = o;
    }
}

// DropPath (vision_transformer.py:27-35) for a whole backbone pass: out[blk][j] = Bernoulli(keep[blk]) / keep[blk], one
// value per (block, branch, sample); consumed as the GEMM epilogues' per-sample row scale.
// d_seed (optional, device): added to `seed` at run time, so that a HIP graph of the step draws new masks at every replay
__global__ __launch_bounds__(256) void droppath_scales_kernel(const float* __restrict__ keep, float* __restrict__ out,
                                                              int per_block, int nblocks, unsigned long long seed,
                                                              const unsigned long long* __restrict__ d_seed) {
    if (d_seed) seed += *d_seed;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= per_block * nblocks) return;
    const float k = keep[i / per_block];
    if (k >= 1.0f) { out[i] = 1.0f; return; }
    const double t = (1.0 - (double)k) * 4294967296.0;
    const unsigned thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    out[i] = drop_keep(seed, (unsigned long long)i, thr) ? 1.0f / k : 0.f;
}

// ---------------------------------------------------------------------------------------------- embedding
// x[r, :] = emb[tok[r], :] + pos[r % T, :]   then dropout;  one wave per row, D % 4 == 0
__global__ __launch_bounds__(256) void dec_embed_fwd_kernel(const long long* __restrict__ tok, const float* __restrict__ emb,
                                                            const float* __restrict__ pos, float* __restrict__ x, int rows,
                                                            int T, int D, int num_classes, unsigned long long seed,
                                                            unsigned thr, float scale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    long long c = tok[r];
    if (c < 0 || c >= num_classes) c = 0;                              // torch raises; never index out of the table
    const float* e = emb + c * D;
    const float* p = pos + (long)(r % T) * D;
    for (int d = lane * 4; d < D; d += 256) {
        f32x4v a = *reinterpret_cast<const f32x4v*>(e + d);
        const f32x4v b = *reinterpret_cast<const f32x4v*>(p + d);
        float v[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = (thr == 0u || drop_keep(seed, (unsigned long long)((long)r * D + d + k), thr)) ? v[k] * scale : 0.f;
        a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3];
        *reinterpret_cast<f32x4v*>(x + (long)r * D + d) = a;
    }
}
// demb[c, :] += sum over rows with tok == c of keep * scale * dx[r, :]   (c != padding_idx: nn.Embedding(padding_idx)).
// Workgroup (c, chunk) scans rows [chunk * rows_per_block, ...) for class c (ballots; hits summed in row order) and
// publishes its partial row with fp32 atomics - the <BOS/EOS> class collects 2 hits per sample, all others a handful.
// Thread t owns columns t, t+256, ...
__global__ __launch_bounds__(256) void dec_embed_bwd_kernel(const long long* __restrict__ tok, const float* __restrict__ dx,
                                                            float* __restrict__ demb, int rows, int D, int padding_idx,
                                                            unsigned long long seed, unsigned thr, float scale,
                                                            int rows_per_block) {
    __shared__ unsigned long long match[4];
    const int c = blockIdx.x;
    if (c == padding_idx) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                              // D <= 1024
    bool any = false;
    const int row0 = blockIdx.y * rows_per_block;
    const int row1 = row0 + rows_per_block < rows ? row0 + rows_per_block : rows;
    for (int base = row0; base < row1; base += 256) {
        const int r = base + threadIdx.x;
        const unsigned long long m = ballot(r < row1 && tok[r] == c);
        if ((threadIdx.x & 63) == 0) match[threadIdx.x >> 6] = m;
        __syncthreads();
        for (int wv = 0; wv < 4; ++wv) {
            unsigned long long mm = match[wv];
            any = any || mm != 0ull;
            for (int bit = 0; mm; ++bit, mm >>= 1) {
                if (!(mm & 1ull)) continue;
                const int rr = base + 64 * wv + bit;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int d = threadIdx.x + 256 * k;
                    if (d < D) {
                        const float g = dx[(long)rr * D + d];
                        const bool keep = thr == 0u || drop_keep(seed, (unsigned long long)((long)rr * D + d), thr);
                        acc[k] += keep ? g * scale : 0.f;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!any) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = threadIdx.x + 256 * k;
        if (d < D) atomicAdd(demb + (long)c * D + d, acc[k]);
    }
}

// ------------------------------------------------------------------------------------- decoder attention
// One workgroup (256 threads) per (sample, head).  Scores: thread j owns key j (its K row in registers) and loops over
// the <= 32 queries (fp32 Q, pre-scaled by 1/sqrt(d_k), broadcast from LDS); softmax: one wave per query row;
// P.V: thread (d = t & 63, g = t >> 6) owns output column d of queries g, g+4, ...
struct DecAttnParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v;       // rows: q[b*Tq + t], k/v[b*Tk + j]; head h at column 64*h
    long ldq, ldk, ldv;
    bf16_t* out; long ldo;                                   // [B*Tq, H*64]
    float* lse;                                              // [B, H, Tq]   log-sum-exp of the scaled, masked scores
    float* probs;                                            // optional [B, H, Tq, Tk] fp32: softmax AFTER dropout
    const long long* tokens;                                 // optional [B, Tk]: key j masked where tokens == pad_idx
    const int* key_len;                                      // optional [B]: keys j >= key_len[b] masked
    int pad_idx, causal;                                     // causal: key j visible to query t iff j <= t
    int B, H, Tq, Tk;
    float scale;
    unsigned long long seed; unsigned thr; float keep_scale; // dropout on the attention weights
    // backward only
    const bf16_t* d_out; bf16_t* dq; bf16_t* dk; bf16_t* dv; long lddq, lddk, lddv;
};
constexpr int DA_MAXQ = 32, DA_MAXK = 256, DA_D = 64, DA_KSTR = 132;    // K/V LDS rows: 64 bf16 + 4 B pad (33 dwords)
__host__ __device__ inline int dec_attn_fwd_smem(int Tq, int Tk) {
    const int kv = Tk * DA_KSTR, p = Tq * Tk * 4;
    return DA_MAXQ * DA_D * 4 + (kv > p ? kv : p) + kv + 16;
}
__host__ __device__ inline int dec_attn_bwd_smem(int Tq, int Tk) {
    return 2 * DA_MAXQ * DA_D * 4 + Tk * DA_KSTR + Tq * Tk * 4 + 2 * DA_MAXQ * 4 + 16;
}

__device__ __forceinline__ bool dec_attn_visible(const DecAttnParams& p, int b, int t, int j) {
    if (j >= p.Tk) return false;
    if (p.causal && j > t) return false;
    if (p.key_len && j >= p.key_len[b]) return false;
    if (p.tokens && p.tokens[(long)b * p.Tk + j] == (long long)p.pad_idx) return false;
    return true;
}
// fp32 [32][64] image of a [T, 64] bf16 slab (rows >= T zero), values multiplied by mul
__device__ __forceinline__ void dec_attn_stage_f32(const bf16_t* __restrict__ src, long ld, int T, float mul, float* dst) {
    const int row = threadIdx.x >> 3, c = (threadIdx.x & 7) * 8;
    float v[8];
    if (row < T) {
        unpack8(*reinterpret_cast<const u32x4*>(src + (long)row * ld + c), v);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[row * DA_D + c + e] = v[e] * mul;
}
// bf16 rows into LDS with the padded stride
__device__ __forceinline__ void dec_attn_stage_rows(const bf16_t* __restrict__ src, long ld, int T, char* dst) {
    for (int id = threadIdx.x; id < T * 8; id += 256) {
        const int row = id >> 3, slot = id & 7;
        const u32x4 w = *reinterpret_cast<const u32x4*>(src + (long)row * ld + slot * 8);
        unsigned* o = reinterpret_cast<unsigned*>(dst + row * DA_KSTR + slot * 16);
        o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
    }
}
__device__ __forceinline__ void dec_attn_row_regs(const char* img, int row, float* r) {
    const unsigned* s = reinterpret_cast<const unsigned*>(img + row * DA_KSTR);
#pragma unroll
    for (int i = 0; i < 32; ++i) { const unsigned w = s[i]; r[2 * i] = bf_lo(w); r[2 * i + 1] = bf_hi(w); }
}

__global__ __launch_bounds__(256) void dec_attn_fwd_kernel(DecAttnParams p) {
    char* smem = dynamic_smem();
    const int Tq = p.Tq, Tk = p.Tk;
    const int kvb = Tk * DA_KSTR, pb = Tq * Tk * 4;
    float* qs = reinterpret_cast<float*>(smem);                       // [32][64] scaled queries
    char* k_img = smem + DA_MAXQ * DA_D * 4;
    float* pm = reinterpret_cast<float*>(k_img);                      // [Tq][Tk] scores / weights, overlays the K rows
    char* v_img = k_img + (((kvb > pb ? kvb : pb) + 15) & ~15);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    dec_attn_stage_f32(p.q + (long)b * Tq * p.ldq + h * DA_D, p.ldq, Tq, p.scale, qs);
    dec_attn_stage_rows(p.k + (long)b * Tk * p.ldk + h * DA_D, p.ldk, Tk, k_img);
    dec_attn_stage_rows(p.v + (long)b * Tk * p.ldv + h * DA_D, p.ldv, Tk, v_img);
    __syncthreads();
    float s[DA_MAXQ];
    {
        const int j = t;
        float kr[DA_D];
        if (j < Tk) dec_attn_row_regs(k_img, j, kr);
#pragma unroll
        for (int q = 0; q < DA_MAXQ; ++q) {
            float a = 0.f;
            if (q < Tq && j < Tk) {
#pragma unroll
                for (int d = 0; d < DA_D; d += 4) {
                    const f32x4v qv = *reinterpret_cast<const f32x4v*>(qs + q * DA_D + d);
                    a = fmaf(qv.x, kr[d], a); a = fmaf(qv.y, kr[d + 1], a);
                    a = fmaf(qv.z, kr[d + 2], a); a = fmaf(qv.w, kr[d + 3], a);
                }
            }
            s[q] = a;
        }
    }
    __syncthreads();                                                  // every thread is done with the K rows
    if (t < Tk) {
#pragma unroll
        for (int q = 0; q < DA_MAXQ; ++q)
            if (q < Tq) pm[q * Tk + t] = dec_attn_visible(p, b, q, t) ? s[q] : -INFINITY;
    }
    __syncthreads();
    for (int q = w; q < Tq; q += 4) {                                 // softmax of row q by wave w
        float v[4], mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            v[i] = j < Tk ? pm[q * Tk + j] : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = mx == -INFINITY ? 0.f : expf(v[i] - mx); sum += v[i]; }
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        const long prow = (((long)b * p.H + h) * Tq + q) * Tk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            if (j < Tk) {
                float pr = v[i] * inv;
                if (p.thr) pr = drop_keep(p.seed, (unsigned long long)(prow + j), p.thr) ? pr * p.keep_scale : 0.f;
                pm[q * Tk + j] = pr;
                if (p.probs) p.probs[prow + j] = pr;
            }
        }
        if (lane == 0) p.lse[((long)b * p.H + h) * Tq + q] = sum > 0.f ? mx + logf(sum) : 0.f;
    }
    __syncthreads();
    {
        const int d = lane, g = w;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int j = 0; j < Tk; ++j) {
            const float vv = bf2f(*reinterpret_cast<const bf16_t*>(v_img + j * DA_KSTR + d * 2));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = g + 4 * i;
                if (q < Tq) acc[i] = fmaf(pm[q * Tk + j], vv, acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = g + 4 * i;
            if (q < Tq) p.out[((long)b * Tq + q) * p.ldo + h * DA_D + d] = f2bf(acc[i]);
        }
    }
}

// backward: dq, dk, dv of one (sample, head); weights recomputed from the saved log-sum-exp
__global__ __launch_bounds__(256) void dec_attn_bwd_kernel(DecAttnParams p) {
    char* smem = dynamic_smem();
    const int Tq = p.Tq, Tk = p.Tk;
    float* qs = reinterpret_cast<float*>(smem);                       // [32][64] scaled queries
    float* dos = qs + DA_MAXQ * DA_D;                                 // [32][64] d_out
    float* delta = dos + DA_MAXQ * DA_D;                              // [32]
    float* lse = delta + DA_MAXQ;                                     // [32]
    char* k_img = reinterpret_cast<char*>(lse + DA_MAXQ);
    float* ds = reinterpret_cast<float*>(k_img + ((Tk * DA_KSTR + 15) & ~15));   // [Tq][Tk]
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    dec_attn_stage_f32(p.q + (long)b * Tq * p.ldq + h * DA_D, p.ldq, Tq, p.scale, qs);
    dec_attn_stage_f32(p.d_out + (long)b * Tq * p.ldo + h * DA_D, p.ldo, Tq, 1.0f, dos);
    dec_attn_stage_rows(p.k + (long)b * Tk * p.ldk + h * DA_D, p.ldk, Tk, k_img);
    __syncthreads();
    for (int q = w; q < DA_MAXQ; q += 4) {                            // delta[q] = <d_out[q], out[q]>
        float a = 0.f;
        if (q < Tq) a = dos[q * DA_D + lane] * bf2f(p.out[((long)b * Tq + q) * p.ldo + h * DA_D + lane]);
        a = wave_sum(a);
        if (lane == 0) {
            delta[q] = a;
            lse[q] = q < Tq ? p.lse[((long)b * p.H + h) * Tq + q] : 0.f;
        }
    }
    __syncthreads();
    const int j = t;
    if (j < Tk) {
        float pr[DA_MAXQ], dp[DA_MAXQ];
        {
            float kr[DA_D];
            dec_attn_row_regs(k_img, j, kr);
#pragma unroll
            for (int q = 0; q < DA_MAXQ; ++q) {
                float a = 0.f;
                if (q < Tq) {
#pragma unroll
                    for (int d = 0; d < DA_D; d += 4) {
                        const f32x4v qv = *reinterpret_cast<const f32x4v*>(qs + q * DA_D + d);
                        a = fmaf(qv.x, kr[d], a); a = fmaf(qv.y, kr[d + 1], a);
                        a = fmaf(qv.z, kr[d + 2], a); a = fmaf(qv.w, kr[d + 3], a);
                    }
                }
                pr[q] = (q < Tq && dec_attn_visible(p, b, q, j)) ? expf(a - lse[q]) : 0.f;
            }
        }
        float vr[DA_D];
        {
            const bf16_t* vrow = p.v + ((long)b * Tk + j) * p.ldv + h * DA_D;
#pragma unroll
            for (int c = 0; c < DA_D; c += 8) unpack8(*reinterpret_cast<const u32x4*>(vrow + c), vr + c);
        }
        const long pbase = ((long)b * p.H + h) * Tq;
#pragma unroll
        for (int q = 0; q < DA_MAXQ; ++q) {
            float a = 0.f;
            if (q < Tq) {
#pragma unroll
                for (int d = 0; d < DA_D; d += 4) {
                    const f32x4v g = *reinterpret_cast<const f32x4v*>(dos + q * DA_D + d);
                    a = fmaf(g.x, vr[d], a); a = fmaf(g.y, vr[d + 1], a);
                    a = fmaf(g.z, vr[d + 2], a); a = fmaf(g.w, vr[d + 3], a);
                }
            }
            const bool keep = q < Tq && (p.thr == 0u || drop_keep(p.seed, (unsigned long long)((pbase + q) * Tk + j), p.thr));
            dp[q] = keep ? a * p.keep_scale : 0.f;                    // d loss / d (weight before dropout)
        }
        // dV[j, :] = sum_q dropped_weight[q, j] * d_out[q, :]
#pragma unroll
        for (int d = 0; d < DA_D; ++d) vr[d] = 0.f;
#pragma unroll
        for (int q = 0; q < DA_MAXQ; ++q) {
            if (q < Tq) {
                const bool keep = p.thr == 0u || drop_keep(p.seed, (unsigned long long)((pbase + q) * Tk + j), p.thr);
                const float pd = keep ? pr[q] * p.keep_scale : 0.f;
#pragma unroll
                for (int d = 0; d < DA_D; d += 4) {
                    const f32x4v g = *reinterpret_cast<const f32x4v*>(dos + q * DA_D + d);
                    vr[d] = fmaf(pd, g.x, vr[d]); vr[d + 1] = fmaf(pd, g.y, vr[d + 1]);
                    vr[d + 2] = fmaf(pd, g.z, vr[d + 2]); vr[d + 3] = fmaf(pd, g.w, vr[d + 3]);
                }
            }
        }
        {
            bf16_t* o = p.dv + ((long)b * Tk + j) * p.lddv + h * DA_D;
#pragma unroll
            for (int c = 0; c < DA_D; c += 8) *reinterpret_cast<u32x4*>(o + c) = pack8(vr + c);
        }
        // dS = P * (dP - delta);   dK[j, :] = sum_q dS[q, j] * (scale * Q[q, :])
#pragma unroll
        for (int d = 0; d < DA_D; ++d) vr[d] = 0.f;
#pragma unroll
        for (int q = 0; q < DA_MAXQ; ++q) {
            if (q < Tq) {
                const float dsv = pr[q] * (dp[q] - delta[q]);
                ds[q * Tk + j] = dsv;
#pragma unroll
                for (int d = 0; d < DA_D; d += 4) {
                    const f32x4v qv = *reinterpret_cast<const f32x4v*>(qs + q * DA_D + d);
                    vr[d] = fmaf(dsv, qv.x, vr[d]); vr[d + 1] = fmaf(dsv, qv.y, vr[d + 1]);
                    vr[d + 2] = fmaf(dsv, qv.z, vr[d + 2]); vr[d + 3] = fmaf(dsv, qv.w, vr[d + 3]);
                }
            }
        }
        {
            bf16_t* o = p.dk + ((long)b * Tk + j) * p.lddk + h * DA_D;
#pragma unroll
            for (int c = 0; c < DA_D; c += 8) *reinterpret_cast<u32x4*>(o + c) = pack8(vr + c);
        }
    }
    __syncthreads();
    {   // dQ[q, d] = scale * sum_j dS[q, j] * K[j, d]
        const int d = lane, g = w;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int jj = 0; jj < Tk; ++jj) {
            const float kv = bf2f(*reinterpret_cast<const bf16_t*>(k_img + jj * DA_KSTR + d * 2));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = g + 4 * i;
                if (q < Tq) acc[i] = fmaf(ds[q * Tk + jj], kv, acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = g + 4 * i;
            if (q < Tq) p.dq[((long)b * Tq + q) * p.lddq + h * DA_D + d] = f2bf(acc[i] * p.scale);
        }
    }
}

// --------------------------------------------------------------------------------------------------- TFLoss
// row r = (b, t): counted iff t < T-1 and targets[b, t+1] != pad.  acc[0] += -log softmax(logits[r])[target], acc[1] += 1;
// row_lse[r] saved.  One wave per row, C <= 128 classes.
__global__ __launch_bounds__(256) void tf_loss_fwd_kernel(const float* __restrict__ logits, long ldl, int C,
                                                          const long long* __restrict__ targets, int rows, int T,
                                                          int pad_idx, float* __restrict__ row_lse,
                                                          float* __restrict__ acc) {
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float nll = 0.f, cnt = 0.f;                                       // lane 0 of each wave accumulates its rows
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const int t = r % T;
        const float a = lane < C ? logits[(long)r * ldl + lane] : -INFINITY;
        const float bq = lane + 64 < C ? logits[(long)r * ldl + lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(a, bq));
        const float sum = wave_sum((lane < C ? expf(a - mx) : 0.f) + (lane + 64 < C ? expf(bq - mx) : 0.f));
        const float lse = mx + logf(sum);
        if (lane == 0) {
            row_lse[r] = lse;
            if (t < T - 1) {
                const long long tgt = targets[r + 1];
                if (tgt != pad_idx && tgt >= 0 && tgt < C) {
                    nll += lse - logits[(long)r * ldl + tgt];
                    cnt += 1.0f;
                }
            }
        }
    }
    if (lane == 0) { red[w] = nll; red[4 + w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {                                           // one pair of atomics per workgroup
        atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
        atomicAdd(acc + 1, red[4] + red[5] + red[6] + red[7]);
    }
}
// d_logits[r, c] (bf16, ldd columns, zero beyond C and on uncounted rows) = (softmax - onehot) * upstream / count
__global__ __launch_bounds__(256) void tf_loss_bwd_kernel(const float* __restrict__ logits, long ldl, int C,
                                                          const long long* __restrict__ targets, int rows, int T,
                                                          int pad_idx, const float* __restrict__ row_lse,
                                                          const float* __restrict__ acc, const float* __restrict__ upstream,
                                                          bf16_t* __restrict__ d_logits, long ldd) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int t = r % T;
    long long tgt = -1;
    if (t < T - 1) {
        tgt = targets[r + 1];
        if (tgt == pad_idx || tgt < 0 || tgt >= C) tgt = -1;
    }
    const float cnt = acc[1];
    const float sc = (tgt >= 0 && cnt > 0.f) ? (upstream ? upstream[0] : 1.0f) / cnt : 0.f;
    const float lse = row_lse[r];
    for (int c = lane; c < ldd; c += 64) {
        float g = 0.f;
        if (c < C && tgt >= 0) g = (expf(logits[(long)r * ldl + c] - lse) - (c == tgt ? 1.0f : 0.f)) * sc;
        d_logits[(long)r * ldd + c] = f2bf(g);
    }
}

// one decoding position (nrtr_decoder.py:160-168): probs[b, step, :] = softmax(logits[b, :C]); seq[b, step+1] = argmax
__global__ __launch_bounds__(256) void greedy_step_kernel(const float* __restrict__ logits, long ldl, int C, int B,
                                                          float* __restrict__ probs, int steps, int step,
                                                          long long* __restrict__ seq, int seq_len) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float a = lane < C ? logits[(long)b * ldl + lane] : -INFINITY;
    const float bq = lane + 64 < C ? logits[(long)b * ldl + lane + 64] : -INFINITY;
    const float mx = wave_max(fmaxf(a, bq));
    const float ea = lane < C ? expf(a - mx) : 0.f, eb = lane + 64 < C ? expf(bq - mx) : 0.f;
    const float inv = 1.0f / wave_sum(ea + eb);
    float* o = probs + ((long)b * steps + step) * C;
    if (lane < C) o[lane] = ea * inv;
    if (lane + 64 < C) o[lane + 64] = eb * inv;
    // first index of the maximum (torch.max returns the first occurrence on ties)
    int idx = 0x7fffffff;
    if (lane < C && a == mx) idx = lane;
    else if (lane + 64 < C && bq == mx) idx = lane + 64;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int o2 = shfl_xor(idx, m); idx = o2 < idx ? o2 : idx; }
    if (lane == 0 && step + 1 < seq_len) seq[(long)b * seq_len + step + 1] = idx;
}

}  // namespace ccd
