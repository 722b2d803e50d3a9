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
constexpr int AUG_NP = 96;        // floats per (sample, view 1 | view 2): see ccd_amd/dataset/augment.py for the sampler
// pointwise chain (aug_colour):
// p[0] invert (0/1)  p[1] gray alpha  p[2] channel permutation id 0..5  p[3] gamma  p[4..6] per-channel gain
// p[7] contrast alpha (around 128)  p[8] add  p[9] gaussian sigma  p[10] multiplicative noise half range
// p[11] impulse probability  p[12] solarize threshold (>= 256: off)  p[13] noise seed (integer valued)
// neighbourhood members (augment_spatial_kernel, a pre-pass that stages one uint8 image per (sample, view)):
// p[25] JPEG quality 1..100 (0: off) - first, like the `arithmetic` group it belongs to (augmentation_pipelines.py:140)
// p[14] 0 none | 1 7x7 correlation, coefficients p[32 + 7 (dy + 3) + (dx + 3)], BORDER_REFLECT_101 (Gaussian / average / motion
//       blur, Sharpen, Emboss, EdgeDetect) | 2 median, k = p[15] in {3, 5, 7}, replicated border | 3 bilateral, d = p[15],
//       sigma_color = p[26], sigma_space = p[27], BORDER_REFLECT_101            (the `Blur` group, :165-176)
constexpr int AUG_P_MODE = 14, AUG_P_K = 15, AUG_P_JPEG = 25, AUG_P_SIGC = 26, AUG_P_SIGS = 27, AUG_P_KERN = 32;
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


// ---- neighbourhood members: oracle/datapipe_np.py restates each (jpeg_roundtrip pinned against PIL / libjpeg, median_blur and
// filter7 against scipy.ndimage, bilateral_blur after OpenCV's documented algorithm)
__device__ __forceinline__ int aug_reflect101(int i, int n) {
    if (n == 1) return 0;
    i = i < 0 ? -i : i;
    const int period = 2 * (n - 1);
    i %= period;
    return i > n - 1 ? period - i : i;
}
__device__ __forceinline__ int aug_clampi(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }
__device__ __forceinline__ unsigned char aug_round_u8(float v) {
    v = floorf(v + 0.5f);
    return (unsigned char)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
}
// Annex-K base tables (libjpeg's std_luminance_quant_tbl / std_chrominance_quant_tbl), natural order
__device__ __forceinline__ int aug_jpeg_base(bool luma, int i) {
    static constexpr unsigned char L[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                            14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113,
                                            92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static constexpr unsigned char C[32] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                            47, 66, 99, 99, 99, 99, 99, 99};
    return luma ? L[i] : (i < 32 ? C[i] : 99);
}
__host__ __device__ inline int aug_pad16(int n) { return (n + 15) / 16 * 16; }
// dynamic LDS of the pre-pass: the uint8 image + (JPEG) sample planes and coefficient planes of the padded image
__host__ __device__ inline long aug_spatial_smem(int H, int W) {
    const long plane = (long)aug_pad16(H) * aug_pad16(W);
    return (((long)H * W * 3 + 15) / 16) * 16 + 2 * (plane + plane / 2) * 4 + 64 * 4;
}
// one workgroup per (sample, view): img uint8 [B, H, W, 3], params fp32 [B, 2, AUG_NP] -> staged uint8 [B, 2, H, W, 3]
__global__ __launch_bounds__(256) void augment_spatial_kernel(const unsigned char* __restrict__ img, const float* __restrict__ params,
                                                              unsigned char* __restrict__ staged, int H, int W) {
    char* smem = dynamic_smem();
    const int t = threadIdx.x, b = blockIdx.x >> 1, npix = H * W;
    const float* p = params + (long)blockIdx.x * AUG_NP;
    unsigned char* cur = reinterpret_cast<unsigned char*>(smem);
    const int Hp = aug_pad16(H), Wp = aug_pad16(W), Hc = Hp / 2, Wc = Wp / 2;
    float* plane = reinterpret_cast<float*>(smem + ((npix * 3 + 15) / 16) * 16);      // Y [Hp][Wp], Cb [Hc][Wc], Cr [Hc][Wc]
    float* coef = plane + Hp * Wp + 2 * Hc * Wc;
    float* dct = coef + Hp * Wp + 2 * Hc * Wc;                                          // D[u][x]
    const unsigned char* src = img + (long)b * npix * 3;
    for (int i = t; i < npix * 3; i += 256) cur[i] = src[i];
    const int quality = (int)p[AUG_P_JPEG];
    if (quality > 0) {
        if (t < 64) {
            const int u = t >> 3, x = t & 7;
            dct[t] = (u == 0 ? 0.35355339059327379f : 0.5f) * cosf((float)((2 * x + 1) * u) * 0.19634954084936207f);
        }
        __syncthreads();
        // colour conversion (libjpeg's 16-bit fixed point), edges replicated to whole MCUs, chroma as h2v2 averages
        for (int i = t; i < Hp * Wp; i += 256) {
            const int y = i / Wp, x = i % Wp, sp = (aug_clampi(y, H) * W + aug_clampi(x, W)) * 3;
            const int R = cur[sp], G = cur[sp + 1], Bl = cur[sp + 2];
            plane[i] = (float)((19595 * R + 38470 * G + 7471 * Bl + 32768) >> 16) - 128.f;
        }
        for (int i = t; i < Hc * Wc; i += 256) {
            const int cy = i / Wc, cx = i % Wc;
            int sb = 0, sr = 0;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int sp = (aug_clampi(2 * cy + dy, H) * W + aug_clampi(2 * cx + dx, W)) * 3;
                    const int R = cur[sp], G = cur[sp + 1], Bl = cur[sp + 2];
                    sb += (-11059 * R - 21709 * G + 32768 * Bl + 8421375) >> 16;
                    sr += (32768 * R - 27439 * G - 5329 * Bl + 8421375) >> 16;
                }
            const int bias = (cx & 1) ? 2 : 1;
            plane[Hp * Wp + i] = (float)((sb + bias) >> 2) - 128.f;
            plane[Hp * Wp + Hc * Wc + i] = (float)((sr + bias) >> 2) - 128.f;
        }
        __syncthreads();
        const int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
        const int scale = q < 50 ? 5000 / q : 200 - 2 * q;
        // forward DCT + quantisation (round half away from zero) + dequantisation, one coefficient per thread and step
        const int ntot = Hp * Wp + 2 * Hc * Wc;
        for (int i = t; i < ntot; i += 256) {
            const bool luma = i < Hp * Wp;
            const int j = luma ? i : (i - Hp * Wp) % (Hc * Wc), base = luma ? 0 : i - j, pw = luma ? Wp : Wc;
            const int y = j / pw, x = j % pw, u = y & 7, v = x & 7;
            const float* blk = plane + base + (y - u) * pw + (x - v);
            float acc = 0.f;
#pragma unroll
            for (int yy = 0; yy < 8; ++yy) {
                float row = 0.f;
#pragma unroll
                for (int xx = 0; xx < 8; ++xx) row = fmaf(blk[yy * pw + xx], dct[v * 8 + xx], row);
                acc = fmaf(dct[u * 8 + yy], row, acc);
            }
            int tq = (aug_jpeg_base(luma, u * 8 + v) * scale + 50) / 100;
            tq = tq < 1 ? 1 : (tq > 255 ? 255 : tq);
            const float lvl = floorf(fabsf(acc) / (float)tq + 0.5f);
            coef[i] = (acc < 0.f ? -lvl : lvl) * (float)tq;
        }
        __syncthreads();
        // inverse DCT, level shift, clamp to 0..255 (integer valued)
        for (int i = t; i < ntot; i += 256) {
            const bool luma = i < Hp * Wp;
            const int j = luma ? i : (i - Hp * Wp) % (Hc * Wc), base = luma ? 0 : i - j, pw = luma ? Wp : Wc;
            const int y = j / pw, x = j % pw, yy = y & 7, xx = x & 7;
            const float* blk = coef + base + (y - yy) * pw + (x - xx);
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float row = 0.f;
#pragma unroll
                for (int v = 0; v < 8; ++v) row = fmaf(blk[u * pw + v], dct[v * 8 + xx], row);
                acc = fmaf(dct[u * 8 + yy], row, acc);
            }
            const float v = floorf(acc + 128.5f);
            plane[i] = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        }
        __syncthreads();
        // "fancy" chroma up-sampling (3/4 nearer + 1/4 further in each direction) + YCbCr -> RGB, back into the uint8 image
        for (int i = t; i < npix; i += 256) {
            const int y = i / W, x = i % W, cy = y >> 1, cx = x >> 1;
            const int fy = aug_clampi((y & 1) ? cy + 1 : cy - 1, Hc), nx = aug_clampi((x & 1) ? cx + 1 : cx - 1, Wc);
            int ch[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float* P = plane + Hp * Wp + k * Hc * Wc;
                const int col = 3 * (int)P[cy * Wc + cx] + (int)P[fy * Wc + cx], nb = 3 * (int)P[cy * Wc + nx] + (int)P[fy * Wc + nx];
                ch[k] = ((3 * col + nb + ((x & 1) ? 7 : 8)) >> 4) - 128;
            }
            const int Y = (int)plane[y * Wp + x];
            const int r = Y + ((91881 * ch[1] + 32768) >> 16);
            const int g = Y + ((-22554 * ch[0] - 46802 * ch[1] + 32768) >> 16);
            const int bl = Y + ((116130 * ch[0] + 32768) >> 16);
            cur[i * 3] = (unsigned char)aug_clampi(r, 256);
            cur[i * 3 + 1] = (unsigned char)aug_clampi(g, 256);
            cur[i * 3 + 2] = (unsigned char)aug_clampi(bl, 256);
        }
    }
    __syncthreads();
    unsigned char* dst = staged + (long)blockIdx.x * npix * 3;
    const int mode = (int)p[AUG_P_MODE];
    for (int i = t; i < npix; i += 256) {
        const int y = i / W, x = i % W;
        if (mode == 1) {                                     // 7 x 7 correlation
            float acc[3] = {0.f, 0.f, 0.f};
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -3; dx <= 3; ++dx) {
                    const float wgt = p[AUG_P_KERN + 7 * (dy + 3) + (dx + 3)];
                    if (wgt == 0.f) continue;
                    const int sp = (aug_reflect101(y + dy, H) * W + aug_reflect101(x + dx, W)) * 3;
                    acc[0] = fmaf(wgt, (float)cur[sp], acc[0]); acc[1] = fmaf(wgt, (float)cur[sp + 1], acc[1]);
                    acc[2] = fmaf(wgt, (float)cur[sp + 2], acc[2]);
                }
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[i * 3 + k] = aug_round_u8(acc[k]);
        } else if (mode == 2) {                              // median of the k x k window: the value of rank (k k) / 2
            const int r = (int)p[AUG_P_K] / 2, n = (2 * r + 1) * (2 * r + 1), want = n / 2;
            for (int k = 0; k < 3; ++k) {
                int med = cur[i * 3 + k];
                for (int a = 0; a < n; ++a) {
                    const int va = cur[(aug_clampi(y + a / (2 * r + 1) - r, H) * W + aug_clampi(x + a % (2 * r + 1) - r, W)) * 3 + k];
                    int less = 0, leq = 0;
                    for (int c = 0; c < n; ++c) {
                        const int vc = cur[(aug_clampi(y + c / (2 * r + 1) - r, H) * W + aug_clampi(x + c % (2 * r + 1) - r, W)) * 3 + k];
                        less += vc < va;
                        leq += vc <= va;
                    }
                    if (less <= want && want < leq) { med = va; break; }
                }
                dst[i * 3 + k] = (unsigned char)med;
            }
        } else if (mode == 3) {                              // bilateral (OpenCV: L1 colour distance, circular window)
            const int radius = (int)p[AUG_P_K] / 2;
            const float gc = -0.5f / (p[AUG_P_SIGC] * p[AUG_P_SIGC]), gs = -0.5f / (p[AUG_P_SIGS] * p[AUG_P_SIGS]);
            const float c0 = cur[i * 3], c1 = cur[i * 3 + 1], c2 = cur[i * 3 + 2];
            float num[3] = {0.f, 0.f, 0.f}, den = 0.f;
            for (int dy = -radius; dy <= radius; ++dy)
                for (int dx = -radius; dx <= radius; ++dx) {
                    const int rr = dy * dy + dx * dx;
                    if (rr > radius * radius) continue;
                    const int sp = (aug_reflect101(y + dy, H) * W + aug_reflect101(x + dx, W)) * 3;
                    const float t0 = cur[sp], t1 = cur[sp + 1], t2 = cur[sp + 2];
                    const float dist = fabsf(t0 - c0) + fabsf(t1 - c1) + fabsf(t2 - c2);
                    const float wgt = expf((float)rr * gs) * expf(dist * dist * gc);
                    num[0] += wgt * t0; num[1] += wgt * t1; num[2] += wgt * t2;
                    den += wgt;
                }
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[i * 3 + k] = aug_round_u8(num[k] / den);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[i * 3 + k] = cur[i * 3 + k];
        }
    }
}

// img uint8 [B, H, W, 3]; staged uint8 [B, 2, H, W, 3] (augment_spatial_kernel: the neighbourhood members of views 1 / 2);
// params fp32 [B, 2, AUG_NP] (view 1, view 2); theta fp32 [B, 3, 3]; out fp32 [B, 3 views, 3, H, W]
__global__ __launch_bounds__(256) void augment_views_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ staged,
                                                            const float* __restrict__ params,
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
        float c[3];
        const unsigned char* s1 = staged + (long)(2 * b) * H * W * 3 + pix * 3;
        aug_colour(params + (long)b * 2 * AUG_NP, (float)s1[0], (float)s1[1], (float)s1[2], (unsigned)pix, c);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[(3 + k) * plane] = (c[k] * (1.0f / 255.f) - mean[k]) * istd[k];
    }
    {
        const float* th = theta + (long)b * 9;
        const float* p2 = params + ((long)b * 2 + 1) * AUG_NP;
        const unsigned char* s2 = staged + (long)(2 * b + 1) * H * W * 3;
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
                    float c[3];
                    aug_colour(p2, (float)s2[sp * 3], (float)s2[sp * 3 + 1], (float)s2[sp * 3 + 2], (unsigned)sp, c);
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[k] += wgt * c[k];
                }
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) o[(6 + k) * plane] = (acc[k] * (1.0f / 255.f) - mean[k]) * istd[k];
    }
}

}  // namespace ccd
