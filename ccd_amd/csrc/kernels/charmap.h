// charmap.h - the character-region path of ABIDINOModel on the device, in the compact "id map" form
// (uint8 per pixel: index of the character plane, 255 = none) instead of 26 dense fp32 planes:
//   ccl_label_kernel        label_cluster.forward                      Dino/utils/DBSCAN.py:65-103
//   warp_idmap_kernel       affine_grid + grid_sample(bilinear) > 0.1   Dino/model/dino_vision.py:72-77, train.py:234-236
//   region_stats_kernel     interpolate(x1/4) + per-plane sums + index  Dino/model/dino_vision.py:38-43,48
//   select_scan_kernel      length clamp + new_index + row offsets      Dino/model/dino_vision.py:82-85
//   region_pool_fwd/bwd     bmm of the normalised maps with the tokens + row gather   :44-47, 87
// Integer work is exact; the only floating-point decisions are the two thresholds of the warp (documented there).
// Two kept components are never 8-adjacent, so every 2x2 block of the LABELLED map - the bilinear footprint of the warp -
// touches at most ONE plane: a warped pixel lies on at most one plane and an id map represents the warped view without
// loss too.  Neighbouring pixels of the warped map may carry different ids though (a one-pixel gap can close under a
// sub-pixel shift or a zoom-out), so the x1/4 down-sampling keeps up to four planes per token (region_stats_kernel).
#pragma once

namespace ccd {

constexpr int CM_H = 32, CM_W = 128, CM_PIX = CM_H * CM_W, CM_PLANES = 26, CM_MIN_AREA = 30;
constexpr unsigned char CM_BG = 255;

// ---- 8-connected component labelling of one 32x128 mask per workgroup (256 threads, 16 pixels each) --------
// label = raster index of the component's first pixel (== skimage/scipy numbering order); components are kept
// in that order while area >= 30, at most 26; planes are then ordered by mean column (exact rational compare,
// ties -> lower label first; the reference's np.argsort is unstable on ties).
__global__ __launch_bounds__(256) void ccl_label_kernel(const float* __restrict__ mask, unsigned char* __restrict__ idmap,
                                                        int images) {
    __shared__ int label[CM_PIX];
    __shared__ int area[CM_PIX];
    __shared__ int colsum[CM_PIX];
    __shared__ int scan[257];
    __shared__ int kept_root[CM_PLANES];
    __shared__ int kept_plane[CM_PLANES];
    __shared__ int changed;
    __shared__ int nkept;
    const int t = threadIdx.x;
    const float* m = mask + (long)blockIdx.x * CM_PIX;
    for (int i = t; i < CM_PIX; i += 256) {
        label[i] = m[i] != 0.0f ? i : -1;
        area[i] = 0;
        colsum[i] = 0;
    }
    __syncthreads();
    for (;;) {
        if (t == 0) changed = 0;
        __syncthreads();
        // hook: pull the smallest neighbouring label onto this pixel's current root
        for (int i = t; i < CM_PIX; i += 256) {
            const int li = label[i];
            if (li < 0) continue;
            const int y = i >> 7, x = i & 127;
            int best = li;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= CM_H || xx < 0 || xx >= CM_W) continue;
                    const int ln = label[yy * CM_W + xx];
                    if (ln >= 0 && ln < best) best = ln;
                }
            if (best < li) {
                atomicMin(&label[li], best);
                atomicMin(&label[i], best);
                changed = 1;
            }
        }
        __syncthreads();
        // compress: point every pixel at its current root
        for (int i = t; i < CM_PIX; i += 256) {
            int l = label[i];
            if (l < 0) continue;
            while (label[l] != l) l = label[l];
            label[i] = l;
        }
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    for (int i = t; i < CM_PIX; i += 256) {
        const int l = label[i];
        if (l >= 0) {
            atomicAdd(&area[l], 1);
            atomicAdd(&colsum[l], i & 127);
        }
    }
    __syncthreads();
    // ordered selection of the first 26 qualifying roots (raster order): block-wide exclusive scan of flags
    int cnt = 0;
    for (int k = 0; k < 16; ++k) {
        const int i = t * 16 + k;
        cnt += (label[i] == i && area[i] >= CM_MIN_AREA) ? 1 : 0;
    }
    scan[t + 1] = cnt;
    if (t == 0) scan[0] = 0;
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k <= 256; ++k) scan[k] += scan[k - 1];
        nkept = scan[256] < CM_PLANES ? scan[256] : CM_PLANES;
    }
    __syncthreads();
    int rank = scan[t];
    for (int k = 0; k < 16; ++k) {
        const int i = t * 16 + k;
        if (label[i] == i && area[i] >= CM_MIN_AREA) {
            if (rank < CM_PLANES) kept_root[rank] = i;
            ++rank;
        }
    }
    __syncthreads();
    if (t < nkept) {
        const int ra = kept_root[t];
        const long long sa = colsum[ra], aa = area[ra];
        int plane = 0;
        for (int o = 0; o < nkept; ++o) {
            const int rb = kept_root[o];
            const long long lhs = (long long)colsum[rb] * aa, rhs = sa * (long long)area[rb];   // mean_b ? mean_a
            if (lhs < rhs || (lhs == rhs && o < t)) ++plane;
        }
        kept_plane[t] = plane;
    }
    __syncthreads();
    // reuse area[] as root -> plane map (-1 = dropped)
    for (int i = t; i < CM_PIX; i += 256) area[i] = -1;
    __syncthreads();
    if (t < nkept) area[kept_root[t]] = kept_plane[t];
    __syncthreads();
    unsigned char* out = idmap + (long)blockIdx.x * CM_PIX;
    for (int i = t; i < CM_PIX; i += 256) {
        const int l = label[i];
        const int p = l >= 0 ? area[l] : -1;
        out[i] = p >= 0 ? (unsigned char)p : CM_BG;
    }
}

// float mask (0/1) -> id map with plane 0 for text, used to push the dataset mask through warp_idmap_kernel
__global__ void mask_to_idmap_kernel(const float* __restrict__ mask, unsigned char* __restrict__ idmap, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idmap[i] = mask[i] != 0.0f ? 0 : CM_BG;
}
// predicted-mask branch (dino_vision.py:64-66): softmax(seg)[:,1] > 0.5  <=>  logit1 > logit0 (2 classes)
__global__ void seg_to_mask_kernel(const float* __restrict__ seg_logits, float* __restrict__ mask, int images) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)images * CM_PIX) return;
    const long img = i / CM_PIX, pix = i % CM_PIX;
    const float l0 = seg_logits[(img * 2 + 0) * CM_PIX + pix], l1 = seg_logits[(img * 2 + 1) * CM_PIX + pix];
    // exactly softmax's arithmetic: e1 / (e0 + e1) > 0.5 with the max subtracted
    const float mx = fmaxf(l0, l1);
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
    mask[i] = (e1 / (e0 + e1)) > 0.5f ? 1.0f : 0.0f;
}

// ---- view-2 id map: F.affine_grid(theta[:, :2]) + F.grid_sample(bilinear, zeros, align_corners=False) > 0.1 ---
// fp32 arithmetic in ATen's order, no contraction: base grid x_j = (2j+1)/W - 1, g = x*t0 + y*t1 + t2,
// pixel coordinate ((g+1)*size-1)/2, corner weights (x1-ix)(y1-iy) .. accumulated nw, ne, sw, se.
__global__ __launch_bounds__(256) void warp_idmap_kernel(const unsigned char* __restrict__ src,
                                                         const float* __restrict__ theta, int theta_stride,
                                                         unsigned char* __restrict__ dst, int images) {
    const int img = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int y = pix >> 7, x = pix & 127;
    const float* th = theta + (long)img * theta_stride;
    const float xn = __fsub_rn(__fdiv_rn((float)(2 * x + 1), (float)CM_W), 1.0f);
    const float yn = __fsub_rn(__fdiv_rn((float)(2 * y + 1), (float)CM_H), 1.0f);
    const float gx = __fadd_rn(__fadd_rn(__fmul_rn(xn, th[0]), __fmul_rn(yn, th[1])), th[2]);
    const float gy = __fadd_rn(__fadd_rn(__fmul_rn(xn, th[3]), __fmul_rn(yn, th[4])), th[5]);
    const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), (float)CM_W), 1.0f), 2.0f);
    const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), (float)CM_H), 1.0f), 2.0f);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx1 = __fsub_rn(ix, fx), wx0 = __fsub_rn(__fadd_rn(fx, 1.0f), ix);
    const float wy1 = __fsub_rn(iy, fy), wy0 = __fsub_rn(__fadd_rn(fy, 1.0f), iy);
    const float w4[4] = {__fmul_rn(wx0, wy0), __fmul_rn(wx1, wy0), __fmul_rn(wx0, wy1), __fmul_rn(wx1, wy1)};
    const unsigned char* s = src + (long)img * CM_PIX;
    // out-of-range or non-finite coordinates contribute nothing (zeros padding)
    const bool finite = (fx > -1.0e9f && fx < 1.0e9f && fy > -1.0e9f && fy < 1.0e9f);
    const int x0 = finite ? (int)fx : -5, y0 = finite ? (int)fy : -5;
    int id = CM_BG;
    float v = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
        if (xx < 0 || xx >= CM_W || yy < 0 || yy >= CM_H) continue;
        const unsigned char p = s[yy * CM_W + xx];
        if (p == CM_BG) continue;
        id = p;                       // all non-background corners carry the same id (see header)
        v = __fadd_rn(v, w4[c]);
    }
    dst[(long)img * CM_PIX + pix] = (id != CM_BG && v > 0.1f) ? (unsigned char)id : CM_BG;
}

// ---- per view: token -> up to 4 (plane, coefficient) pairs, plane presence -----------------------------------
// bilinear x1/4 with align_corners=False samples exactly the mean of the central 2x2 of each 4x4 cell
// (verified bit-exact), so for plane j  w_t(j) = (#central pixels of token t on plane j) / 4 and
// coef_t(j) = w_t(j) / sum_t' w_t'(j) (fp32 division like the reference's `clusters / max_cluster_index`).
// In the labelled view the central 2x2 touches at most one plane (kept components are never 8-adjacent); in the WARPED
// view two components that were one pixel apart can end up side by side (sub-pixel shift, zoom-out), and the reference
// then credits the token to both planes - hence up to four pairs per token, slot s of token t at [t][s].
__global__ __launch_bounds__(256) void region_stats_kernel(const unsigned char* __restrict__ idmap,
                                                           unsigned char* __restrict__ tok_plane,
                                                           float* __restrict__ tok_coef,
                                                           unsigned char* __restrict__ present, int views) {
    __shared__ float plane_sum[CM_PLANES];
    const int t = threadIdx.x;           // token index, 8 x 32 grid
    const unsigned char* im = idmap + (long)blockIdx.x * CM_PIX;
    if (t < CM_PLANES) plane_sum[t] = 0.0f;
    __syncthreads();
    const int ty = t >> 5, tx = t & 31;
    int id[4] = {CM_BG, CM_BG, CM_BG, CM_BG}, cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int dy = 1; dy <= 2; ++dy)
#pragma unroll
        for (int dx = 1; dx <= 2; ++dx) {
            const int p = im[(4 * ty + dy) * CM_W + 4 * tx + dx];
            if (p == CM_BG) continue;
            bool placed = false;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (placed) continue;
                if (id[s] == p) { ++cnt[s]; placed = true; }
                else if (id[s] == CM_BG) { id[s] = p; cnt[s] = 1; placed = true; }
            }
        }
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (id[s] != CM_BG) atomicAdd(&plane_sum[id[s]], 0.25f * (float)cnt[s]);     // multiples of 0.25: exact in any order
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        tok_plane[((long)blockIdx.x * 256 + t) * 4 + s] = (unsigned char)id[s];
        tok_coef[((long)blockIdx.x * 256 + t) * 4 + s] = id[s] != CM_BG ? (0.25f * (float)cnt[s]) / plane_sum[id[s]] : 0.0f;
    }
    if (t < CM_PLANES) present[(long)blockIdx.x * CM_PLANES + t] = plane_sum[t] > 0.0f ? 1 : 0;
}

// ---- one block: rows kept per image (from view 1), row offsets, total -------------------------------------
// sel[0] = M (rows per view), new_index[b][j] = j <= clamp(#present planes of view-1 image b, 3, 26)
__global__ __launch_bounds__(256) void select_scan_kernel(const unsigned char* __restrict__ present, int batch,
                                                          int* __restrict__ nsel, int* __restrict__ offset,
                                                          int* __restrict__ total,
                                                          unsigned char* __restrict__ new_index) {
    __shared__ int part[256];
    const int t = threadIdx.x;
    const int per = (batch + 255) / 256;
    int local = 0;
    for (int k = 0; k < per; ++k) {
        const int b = t * per + k;
        if (b >= batch) break;
        int c = 0;
        for (int j = 0; j < CM_PLANES; ++j) c += present[(long)b * CM_PLANES + j];
        c = c < 3 ? 3 : (c > 26 ? 26 : c);
        const int n = c + 1 < CM_PLANES ? c + 1 : CM_PLANES;
        nsel[b] = n;
        local += n;
        for (int j = 0; j < CM_PLANES; ++j) new_index[(long)b * CM_PLANES + j] = j <= c ? 1 : 0;
    }
    part[t] = local;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int k = 0; k < 256; ++k) { const int v = part[k]; part[k] = run; run += v; }
        total[0] = run;
    }
    __syncthreads();
    int run = part[t];
    for (int k = 0; k < per; ++k) {
        const int b = t * per + k;
        if (b >= batch) break;
        offset[b] = run;
        run += nsel[b];
    }
}

// ---- masked region pooling + row gather: rows[half*M + off_b + j, :] = sum over pairs (t, s) with plane_t(s) == j of coef_t(s) * feat[t, :]
__global__ __launch_bounds__(256) void region_pool_fwd_kernel(const bf16_t* __restrict__ feat,
                                                              const unsigned char* __restrict__ tok_plane,
                                                              const float* __restrict__ tok_coef,
                                                              const int* __restrict__ nsel, const int* __restrict__ offset,
                                                              const int* __restrict__ total, bf16_t* __restrict__ rows,
                                                              int batch, int E) {
    float* acc = reinterpret_cast<float*>(dynamic_smem());      // [26][E]
    __shared__ unsigned char s_plane[256 * 4];
    __shared__ float s_coef[256 * 4];
    const int view = blockIdx.x, b = view % batch, half = view / batch;
    const int t = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        s_plane[4 * t + s] = tok_plane[((long)view * 256 + t) * 4 + s];
        s_coef[4 * t + s] = tok_coef[((long)view * 256 + t) * 4 + s];
    }
    const int n = nsel[b];
    for (int i = t; i < n * E; i += 256) acc[i] = 0.0f;
    __syncthreads();
    const bf16_t* f = feat + (long)view * 256 * E;
    // a thread owns two adjacent channels (its cells of `acc`: no atomics) and walks the tokens eight at a time - eight loads in
    // flight instead of one 2-byte load per trip (the walk is a chain of LDS read-modify-writes: latency-bound as written first,
    // 150 us for a 100-MB read)
    for (int e2 = t; 2 * e2 < E; e2 += 256) {
        for (int tok0 = 0; tok0 < 256; tok0 += 8) {
            unsigned wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const unsigned*>(f + (long)(tok0 + u) * E + 2 * e2);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tok = tok0 + u;
                if (s_plane[4 * tok] == CM_BG) continue;               // slots fill from 0: no pair at all
                const float x0 = bf_lo(wv[u]), x1 = bf_hi(wv[u]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int p = s_plane[4 * tok + s];
                    if (p < n) {
                        const float cf = s_coef[4 * tok + s];
                        acc[p * E + 2 * e2] += cf * x0;
                        acc[p * E + 2 * e2 + 1] += cf * x1;
                    }
                }
            }
        }
    }
    __syncthreads();
    bf16_t* out = rows + ((long)half * total[0] + offset[b]) * E;
    for (int i = t; i < n * E; i += 256) out[i] = f2bf(acc[i]);
}

// d_feat[t, :] = sum over the token's pairs s with plane_t(s) < nsel of coef_t(s) * d_rows[row(plane_t(s)), :]
__global__ __launch_bounds__(256) void region_pool_bwd_kernel(const bf16_t* __restrict__ d_rows,
                                                              const unsigned char* __restrict__ tok_plane,
                                                              const float* __restrict__ tok_coef,
                                                              const int* __restrict__ nsel, const int* __restrict__ offset,
                                                              const int* __restrict__ total, bf16_t* __restrict__ d_feat,
                                                              int batch, int E) {
    const int view = blockIdx.x, b = view % batch, half = view / batch;
    const int n = nsel[b];
    const bf16_t* base = d_rows + ((long)half * total[0] + offset[b]) * E;
    bf16_t* out = d_feat + (long)view * 256 * E;
    const int e8 = E >> 3;
    for (int i = threadIdx.x; i < 256 * e8; i += 256) {
        const int tok = i / e8, c = i % e8;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int p = tok_plane[((long)view * 256 + tok) * 4 + s];
            if (p < n) {
                const float coef = tok_coef[((long)view * 256 + tok) * 4 + s];
                float v[8];
                unpack8(*reinterpret_cast<const u32x4*>(base + (long)p * E + c * 8), v);
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += v[k] * coef;
            }
        }
        *reinterpret_cast<u32x4*>(out + (long)tok * E + c * 8) = pack8(a);
    }
}

// id map -> 26 dense fp32 planes (only for the reference-compatible 'zero' output of ABIDINOModel)
__global__ void idmap_to_planes_kernel(const unsigned char* __restrict__ idmap, float* __restrict__ planes, long images) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= images * CM_PLANES * CM_PIX) return;
    const long img = i / (CM_PLANES * CM_PIX);
    const int plane = (int)((i / CM_PIX) % CM_PLANES), pix = (int)(i % CM_PIX);
    planes[i] = idmap[img * CM_PIX + pix] == plane ? 1.0f : 0.0f;
}
// dense planes (someone else's clusters) -> id map; pixels on several planes keep the lowest plane index
__global__ void planes_to_idmap_kernel(const float* __restrict__ planes, unsigned char* __restrict__ idmap, long images) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= images * CM_PIX) return;
    const long img = i / CM_PIX;
    const int pix = (int)(i % CM_PIX);
    int id = CM_BG;
    for (int p = CM_PLANES - 1; p >= 0; --p)
        if (planes[(img * CM_PLANES + p) * CM_PIX + pix] > 0.0f) id = p;
    idmap[i] = (unsigned char)id;
}

}  // namespace ccd
