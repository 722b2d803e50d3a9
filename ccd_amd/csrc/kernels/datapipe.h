// datapipe.h - the two device steps of the data pipeline.
//
// (1) kmeans2_mask_kernel: the text mask of a word image, mask_create/generate_mask.py:13-29 (= Dino/utils/kmeans.py:7-23):
//     2-means on the gray values, then "text is the cluster that does NOT own the border".  On one axis 2-means is a
//     threshold; scipy.cluster.vq.kmeans runs Lloyd's iteration from 20 random pairs of pixels and keeps the result
//     with the smallest MEAN (not squared) distance, i.e. the best fixed point of the iteration.  Here: a 256-bin
//     histogram, every threshold tested for being a fixed point, the one with the smallest mean absolute distance kept
//     (fp64, operation order fixed, ties -> lowest threshold) - identical to the reference on all fixtures of
//     tests/golden/kmeans_masks.npz, where the global squared-error optimum differs in 10 of 48.  code = 1 for the brighter
//     cluster; flipped when at least 3 of the 4 border lines are mostly 1 (sum > length // 2, integer compare).
//     One workgroup per image of a ragged batch; integer arithmetic except the 256 scores.
// (2) augment_views_kernel: the three views of a sample (datasetsupervised_kmeans.py:48-87): view 0 plain, view 1
//     colour-augmented, view 2 colour-augmented + affine-warped; ImageNet mean/std normalisation (dataset.py:79-80,
//     TF.normalize :80).  The colour stage is the pointwise family of the reference's imgaug pipelines (invert, grayscale
//     blend, channel shuffle, gamma / linear contrast, brightness and per-channel gains, solarize, additive / multiplicative
//     / impulse noise) behind one optional 3x3 filter (blur / sharpen / emboss / edge members); the warp samples the colour-augmented source with bilinear weights and zero fill, at
//     src = W_^-1 theta W_ (x, y, 1) - the exact inverse of how the dataset derives theta from the pixel matrix (:65-71).
#pragma once

namespace ccd {

constexpr int KM_THREADS = 256;

__global__ __launch_bounds__(KM_THREADS) void kmeans2_mask_kernel(const unsigned char* __restrict__ gray,
                                                                  const long* __restrict__ offs, const int* __restrict__ hw,
                                                                  unsigned char* __restrict__ mask) {
    __shared__ int hist[256];
    __shared__ double score[256];
    __shared__ int border[4];
    __shared__ int best_g;
    const int img = blockIdx.x, t = threadIdx.x;
    const long base = offs[img];
    const int h = hw[2 * img], w = hw[2 * img + 1];
    const int n = h * w;
    hist[t] = 0;
    if (t < 4) border[t] = 0;
    __syncthreads();
    for (int i = t; i < n; i += KM_THREADS) atomicAdd(&hist[gray[base + i]], 1);
    __syncthreads();
    {
        // candidate g: cluster 0 = values <= g.  It is a fixed point of Lloyd's iteration when every present value lies on
        // its own side of the centroids' midpoint; its figure of merit is the MEAN ABSOLUTE distance (what scipy's kmeans
        // compares between its restarts - vq's distances are not squared).
        long n0 = 0, s0 = 0, ntot = 0, stot = 0;
        int next_present = 256;
        for (int v = 0; v < 256; ++v) {
            const long c = hist[v];
            ntot += c;
            stot += c * v;
            if (v <= t) { n0 += c; s0 += c * v; }
            else if (c > 0 && next_present == 256) next_present = v;
        }
        const long n1 = ntot - n0, s1 = stot - s0;
        double sc = -1.0;
        if (hist[t] > 0 && n1 > 0) {
            const double m0 = (double)s0 / (double)n0, m1 = (double)s1 / (double)n1;
            const double mid = (m0 + m1) / 2.0;
            if ((double)t < mid && mid < (double)next_present) {
                double acc = 0.0;
                for (int v = 0; v < 256; ++v) {
                    const double d = (double)v - (v <= t ? m0 : m1);
                    acc += (double)hist[v] * (d < 0.0 ? -d : d);
                }
                sc = acc / (double)ntot;
            }
        }
        score[t] = sc;
    }
    __syncthreads();
    if (t == 0) {
        int bg = -1;
        double bs = 0.0;
        for (int g = 0; g < 256; ++g)
            if (score[g] >= 0.0 && (bg < 0 || score[g] < bs)) { bs = score[g]; bg = g; }
        best_g = bg;
    }
    __syncthreads();
    const int g = best_g;
    if (g < 0) {                                    // no two-cluster fixed point (one gray level only): code 0 everywhere
        for (int i = t; i < n; i += KM_THREADS) mask[base + i] = 0;
        return;
    }
    // border sums of the code: first / last column, first / last row
    for (int y = t; y < h; y += KM_THREADS) {
        if (gray[base + (long)y * w] > g) atomicAdd(&border[0], 1);
        if (gray[base + (long)y * w + w - 1] > g) atomicAdd(&border[1], 1);
    }
    for (int x = t; x < w; x += KM_THREADS) {
        if (gray[base + x] > g) atomicAdd(&border[2], 1);
        if (gray[base + (long)(h - 1) * w + x] > g) atomicAdd(&border[3], 1);
    }
    __syncthreads();
    const int num = (border[2] > w / 2) + (border[3] > w / 2) + (border[0] > h / 2) + (border[1] > h / 2);
    const bool flip = num >= 3;
    for (int i = t; i < n; i += KM_THREADS) {
        const bool one = gray[base + i] > g;
        mask[base + i] = (one != flip) ? 1 : 0;
    }
}

// ---- augmentation ---------------------------------------------------------------------------------------------------
constexpr int AUG_NP = 32;        // floats per (sample, view 1 | view 2): see ccd_amd/dataset/augment.py for the sampler
// p[0] invert (0/1)  p[1] gray alpha  p[2] channel permutation id 0..5  p[3] gamma  p[4..6] per-channel gain
// p[7] contrast alpha (around 128)  p[8] add  p[9] gaussian sigma  p[10] multiplicative noise half range
// p[11] impulse probability  p[12] solarize threshold (>= 256: off)  p[13] noise seed (integer valued)
__device__ __forceinline__ unsigned aug_hash(unsigned a, unsigned b) {
    unsigned z = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
    z ^= z >> 16; z *= 0x85EBCA6Bu; z ^= z >> 13; z *= 0xC2B2AE35u; z ^= z >> 16;
    return z;
}
__device__ __forceinline__ float aug_u01(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }
// colour stage of one source pixel: rgb in 0..255 -> rgb in 0..255 (clamped, not rounded: the reference rounds to uint8
// between augmenters; one rounding at the end of the chain is inside the noise every member adds)
__device__ __forceinline__ void aug_colour(const float* __restrict__ p, float r, float g, float b, unsigned pix_id, float* out) {
    float c[3] = {r, g, b};
    if (p[0] != 0.f) { c[0] = 255.f - c[0]; c[1] = 255.f - c[1]; c[2] = 255.f - c[2]; }
    if (p[12] < 256.f) {
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = c[k] >= p[12] ? 255.f - c[k] : c[k];
    }
    const float gray = 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = c[k] + p[1] * (gray - c[k]);
    const int perm = (int)p[2];
    const int p0 = perm >> 1, rest0 = p0 == 0 ? 1 : 0, rest1 = p0 == 2 ? 1 : 2;
    const int p1 = (perm & 1) ? rest1 : rest0, p2 = (perm & 1) ? rest0 : rest1;
    const float s[3] = {c[p0], c[p1], c[p2]};
    const unsigned seed = (unsigned)p[13];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = s[k];
        if (p[3] != 1.0f) v = 255.f * powf(fmaxf(v, 0.f) * (1.0f / 255.f), p[3]);
        v = v * p[4 + k];
        v = 128.f + p[7] * (v - 128.f) + p[8];
        const unsigned h0 = aug_hash(pix_id * 3u + (unsigned)k, seed);
        if (p[9] > 0.f) {          // Box-Muller
            const unsigned h1 = aug_hash(h0, seed ^ 0xA511E9B3u);
            v += p[9] * sqrtf(-2.0f * logf(fmaxf(aug_u01(h0), 1e-7f))) * cosf(6.2831853f * aug_u01(h1));
        }
        if (p[10] > 0.f) v *= 1.0f + p[10] * (2.0f * aug_u01(aug_hash(h0, seed ^ 0x3C6EF372u)) - 1.0f);
        if (p[11] > 0.f) {
            const float u = aug_u01(aug_hash(h0, seed ^ 0xDAA66D2Bu));
            if (u < p[11]) v = u < 0.5f * p[11] ? 0.f : 255.f;
        }
        out[k] = fminf(fmaxf(v, 0.f), 255.f);
    }
}

// the (optionally 3x3-filtered) source pixel
__device__ __forceinline__ void aug_source(const unsigned char* __restrict__ src, int H, int W, int y, int x,
                                           const float* __restrict__ p, float* rgb) {
    if (p[14] == 0.f) {
        const int sp = (y * W + x) * 3;
        rgb[0] = src[sp]; rgb[1] = src[sp + 1]; rgb[2] = src[sp + 2];
        return;
    }
    rgb[0] = rgb[1] = rgb[2] = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int yy = y + dy, xx = x + dx;
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            const float wgt = p[16 + 3 * (dy + 1) + (dx + 1)];
            const int sp = (yy * W + xx) * 3;
            rgb[0] += wgt * src[sp]; rgb[1] += wgt * src[sp + 1]; rgb[2] += wgt * src[sp + 2];
        }
}

// img uint8 [B, H, W, 3]; params fp32 [B, 2, AUG_NP] (view 1, view 2); theta fp32 [B, 3, 3]; out fp32 [B, 3 views, 3, H, W]
__global__ __launch_bounds__(256) void augment_views_kernel(const unsigned char* __restrict__ img, const float* __restrict__ params,
                                                            const float* __restrict__ theta, float* __restrict__ out,
                                                            int B, int H, int W, float m0, float m1, float m2, float is0,
                                                            float is1, float is2) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int y = pix / W, x = pix % W;
    const unsigned char* src = img + (long)b * H * W * 3;
    const float mean[3] = {m0, m1, m2}, istd[3] = {is0, is1, is2};
    float* o = out + (long)b * 9 * H * W + pix;
    const long plane = (long)H * W;
    const float r = src[pix * 3], g = src[pix * 3 + 1], bl = src[pix * 3 + 2];
    {
        const float c[3] = {r, g, bl};
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k * plane] = (c[k] * (1.0f / 255.f) - mean[k]) * istd[k];
    }
    {
        float c[3], f[3];
        aug_source(src, H, W, y, x, params + (long)b * 2 * AUG_NP, f);
        aug_colour(params + (long)b * 2 * AUG_NP, f[0], f[1], f[2], (unsigned)pix, c);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[(3 + k) * plane] = (c[k] * (1.0f / 255.f) - mean[k]) * istd[k];
    }
    {
        const float* th = theta + (long)b * 9;
        const float* p2 = params + ((long)b * 2 + 1) * AUG_NP;
        const float xn = 2.0f * (float)x / (float)(W - 1) - 1.0f, yn = 2.0f * (float)y / (float)(H - 1) - 1.0f;
        const float xs = ((th[0] * xn + th[1] * yn + th[2]) + 1.0f) * 0.5f * (float)(W - 1);
        const float ys = ((th[3] * xn + th[4] * yn + th[5]) + 1.0f) * 0.5f * (float)(H - 1);
        const float xf = floorf(xs), yf = floorf(ys);
        const int x0 = (int)xf, y0 = (int)yf;
        const float ax = xs - xf, ay = ys - yf;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int xx = x0 + dx, yy = y0 + dy;
                const float wgt = (dx ? ax : 1.0f - ax) * (dy ? ay : 1.0f - ay);
                if (xx >= 0 && xx < W && yy >= 0 && yy < H && wgt != 0.f) {
                    const int sp = yy * W + xx;
                    float c[3], f[3];
                    aug_source(src, H, W, yy, xx, p2, f);
                    aug_colour(p2, f[0], f[1], f[2], (unsigned)sp, c);
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[k] += wgt * c[k];
                }
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) o[(6 + k) * plane] = (acc[k] * (1.0f / 255.f) - mean[k]) * istd[k];
    }
}

}  // namespace ccd
