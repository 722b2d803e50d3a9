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
// (2) augment_spatial_kernel + augment_views_kernel: the three views of a sample (datasetsupervised_kmeans.py:48-87): view 0
//     plain, view 1 augmented, view 2 augmented + affine-warped; ImageNet mean/std normalisation (dataset.py:79-80).  The
//     augmentation is the reference's imgaug chain (augmentation_pipelines.py:120-205): one member of each of `arithmetic`,
//     `color`, `Blur`, `contrast`, in that order, applied by ONE workgroup per (sample, view) to a uint8 image held in LDS and
//     rounded between the groups; the warp samples the augmented image with bilinear weights and zero fill at
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
// One parameter row of AUG_NP floats per (sample, view 1 | view 2); ccd_amd/dataset/augment.py draws them with the member lists
// and probabilities of the reference's imgaug pipelines (augmentation_pipelines.py:120-205; dataset_pretrain.py:79-158).  The
// chain runs in the reference's order, ONE member per group, on a uint8 image that is rounded between the groups like imgaug's:
//   p[0] seed   p[1] leading Invert (finetuning pipeline)
//   group `arithmetic`: p[2] = member, p[3..8] = arguments a0..a5, p[9..17] = its 3 x 3 correlation kernel
//      1 AddElementwise (a0 = R: integers -R..R, a1 = per channel)   2 AdditiveGaussianNoise (a0 = sigma, a1 = per channel)
//      3 AdditiveLaplaceNoise (a0 = scale, a1)   4 AdditivePoissonNoise (a0 = lambda, a1)   5 Multiply (a0..a2 = channel gains)
//      6 MultiplyElementwise (a0 = low, a1 = per channel, a2 = high)   7 Dropout (a0 = p, a1)
//      8 CoarseDropout (a0 = p, a1, a2 x a3 = the low-resolution mask, nearest up-sampling)   9 Dropout2d (a0 = kept channels, bits)
//      10 ImpulseNoise / SaltAndPepper / Salt / Pepper (a0 = p, a1, a2 = 0 both | 1 salt | 2 pepper; replacement 255 * Beta(0.5, 0.5))
//      11 Invert   12 Solarize (a0 = threshold)   13 JpegCompression (a0 = PIL quality)
//      14 Emboss / EdgeDetect / DirectedEdgeDetect (kernel, BORDER_REFLECT_101)
//      15 pillike.FilterEdgeEnhanceMore / FilterContour (kernel, a0 = offset, a1 = scale; PIL copies the border pixels)
//   group `color`: p[18] = member, p[19..23] = arguments; HSV = cv2's 8-bit RGB2HSV / HSV2RGB (H in 0..179)
//      1 H += a0 on uint8, saturating (WithColorspace / ChangeColorspace + WithChannels(0, Add))   2 gains a0 * rgb + a1
//      (MultiplyAndAddToBrightness, approximated in RGB)   3 MultiplyHueAndSaturation (a0, a1; H on imgaug's 0..255 scale,
//      wrapped modulo 255)   4 AddToHueAndSaturation (a0 = hue shift in H units, a1 = saturation shift)   5 Grayscale (a0 = alpha)
//      6 KMeansColorQuantization (a0 = k: Lloyd's iteration on the 8-bit Lab triples)   7 UniformColorQuantization (a0 = colours per channel)   8 channel gains a0..a2 (ChangeColorTemperature)   9 ChannelShuffle (a0 = permutation)
//   group `Blur`: p[24] = 0 none | 1 7x7 correlation p[32..80], BORDER_REFLECT_101 (Gaussian / average / motion blur, Sharpen) |
//      2 median, k = p[25] in {3, 5, 7}, replicated border | 3 bilateral, d = p[25], sigma_color = p[26], sigma_space = p[27]
//   group `contrast`: p[28] = member, p[29], p[30]: 1 GammaContrast   2 LinearContrast (around 127)   3 SigmoidContrast (gain, cutoff)
//      4 LogContrast (gain)   6 AllChannelsHistogramEqualization (cv2.equalizeHist per channel)   5 HistogramEqualization (the same on the L
//      channel of 8-bit Lab)   8 AllChannelsCLAHE (p[29] = clip limit, p[30] = tiles a side: cv2.createCLAHE per channel)   7 CLAHE (on L);
//      imgaug's tables truncate
//   group `weather`: p[81] = number of layers (Fog: 1, Clouds: 1 - 2, Snowflakes / Rain: 1 - 3), p[82] = the first one's index among the launch's overlay
//      planes (fp16 [layers][2][H][W], drawn on the host: ccd_amd/dataset/weather.py), p[83] = blend: 0 cloud / rain, v = trunc(clip((1 - plane0) v +
//      plane0 * plane1)); 1 snow, v = round(max(clip(v + plane0), plane1))
// Members not reproduced keep their share of the draw and leave the image unchanged (the list is in INTEGRATION.md).
constexpr int AUG_NP = 96;
constexpr int AUG_P_SEED = 0, AUG_P_PREINV = 1, AUG_P_A = 2, AUG_P_AK = 9, AUG_P_B = 18, AUG_P_C = 24, AUG_P_D = 28, AUG_P_KERN = 32, AUG_P_W = 81;
__device__ __forceinline__ unsigned aug_hash(unsigned a, unsigned b) {
    unsigned z = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
    z ^= z >> 16; z *= 0x85EBCA6Bu; z ^= z >> 13; z *= 0xC2B2AE35u; z ^= z >> 16;
    return z;
}
__device__ __forceinline__ float aug_u01(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }
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
__device__ __forceinline__ unsigned char aug_trunc_u8(float v) {          // np.clip(table, 0, 255).astype(uint8)
    return (unsigned char)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
}
// cv2 (color_hsv.cpp) RGB2HSV_b: 12-bit fixed point with division tables sdiv[v] = round(255 * 4096 / v), hdiv[d] = round(180 * 4096 / (6 d))
__device__ __forceinline__ void aug_rgb2hsv(int r, int g, int b, int& h, int& s, int& v) {
    v = r > g ? r : g; v = v > b ? v : b;
    int vmin = r < g ? r : g; vmin = vmin < b ? vmin : b;
    const int diff = v - vmin;
    s = v ? (diff * (int)rint(1044480.0 / (double)v) + 2048) >> 12 : 0;
    h = 0;
    if (diff) {
        const int hh = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
        h = (hh * (int)rint(122880.0 / (double)diff) + 2048) >> 12;
        if (h < 0) h += 180;
    }
}
// cv2 HSV2RGB_b: through float, h * 6 / 180 reduced into [0, 6), saturate_cast<uchar>(x * 255) (round half to even)
__device__ __forceinline__ void aug_hsv2rgb(int hi, int si, int vi, int& r, int& g, int& b) {
    const float s = (float)si * (1.0f / 255.f), v = (float)vi * (1.0f / 255.f);
    float fr = v, fg = v, fb = v;
    if (si != 0) {
        float h = (float)hi * (6.0f / 180.f);
        while (h < 0.f) h += 6.f;
        while (h >= 6.f) h -= 6.f;
        int sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        const float tab[4] = {v, v * (1.f - s), v * (1.f - s * h), v * (1.f - s * (1.f - h))};
        // (b, g, r) table indices per sector
        const int ib = sector == 0 ? 1 : sector == 1 ? 1 : sector == 2 ? 3 : sector == 3 ? 0 : sector == 4 ? 0 : 2;
        const int ig = sector == 0 ? 3 : sector == 1 ? 0 : sector == 2 ? 0 : sector == 3 ? 2 : sector == 4 ? 1 : 1;
        const int ir = sector == 0 ? 0 : sector == 1 ? 2 : sector == 2 ? 1 : sector == 3 ? 1 : sector == 4 ? 3 : 0;
        fb = tab[ib]; fg = tab[ig]; fr = tab[ir];
    }
    r = aug_clampi((int)rintf(fr * 255.f), 256); g = aug_clampi((int)rintf(fg * 255.f), 256); b = aug_clampi((int)rintf(fb * 255.f), 256);
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
// dynamic LDS of the chain: two uint8 images (a neighbourhood member reads one and writes the other) + (JPEG) sample and
// coefficient planes of the padded image (the histogram member uses the start of that area)
__host__ __device__ inline long aug_image_bytes(int H, int W) { return (((long)H * W * 3 + 15) / 16) * 16; }
__host__ __device__ inline long aug_spatial_smem(int H, int W) {
    const long plane = (long)aug_pad16(H) * aug_pad16(W);
    long work = 2 * (plane + plane / 2) * 4 + 64 * 4;                 // JPEG planes + DCT table
    const long clahe = 12L * 12 * 256 + 4 * 256 * 4;                   // CLAHE: 144 tile tables + a histogram per wave (same area)
    if (work < clahe) work = clahe;
    return 2 * aug_image_bytes(H, W) + work;
}
// the pointwise members of `arithmetic` on one channel value; (y, x) = the pixel, k = the channel
__device__ __forceinline__ float aug_arith_point(const float* __restrict__ p, int op, float v, int y, int x, int k, int H, int W, unsigned seed) {
    const float a0 = p[AUG_P_A + 1], a1 = p[AUG_P_A + 2], a2 = p[AUG_P_A + 3], a3 = p[AUG_P_A + 4];
    const int ch = a1 != 0.f ? k : 0;                                     // per_channel: one draw per channel, else per pixel
    const unsigned e = (unsigned)((y * W + x) * 3 + ch);
    const unsigned h0 = aug_hash(e, seed ^ 0x51ED270Bu);
    switch (op) {
    case 1: return v + (float)((int)floorf(aug_u01(h0) * (2.f * a0 + 1.f)) - (int)a0);
    case 2: {                                                              // Box-Muller
        const unsigned h1 = aug_hash(h0, seed ^ 0xA511E9B3u);
        return v + a0 * sqrtf(-2.0f * logf(fmaxf(aug_u01(h0), 1e-7f))) * cosf(6.2831853f * aug_u01(h1));
    }
    case 3: {
        const float u = aug_u01(h0) - 0.5f, m = fmaxf(1.f - 2.f * fabsf(u), 1e-7f);
        return v - a0 * (u < 0.f ? -1.f : 1.f) * logf(m);
    }
    case 4: {                                                              // Knuth: multiply uniforms until below exp(-lambda)
        const float lim = expf(-a0);
        float prod = 1.f;
        int n = -1;
        unsigned h = h0;
        do {
            prod *= fmaxf(aug_u01(h), 1e-7f);
            h = aug_hash(h, seed ^ 0x3C6EF372u);
            ++n;
        } while (prod > lim && n < 200);
        return v + (float)n;
    }
    case 5: return v * p[AUG_P_A + 1 + k];
    case 6: return v * (a0 + (a2 - a0) * aug_u01(h0));
    case 7: return aug_u01(h0) < a0 ? 0.f : v;
    case 8: {
        const int rows = (int)a2, cols = (int)a3;
        const int cy = aug_clampi(y * rows / H, rows), cx = aug_clampi(x * cols / W, cols);
        return aug_u01(aug_hash((unsigned)((cy * cols + cx) * 3 + ch), seed ^ 0x6A09E667u)) < a0 ? 0.f : v;
    }
    case 9: return ((int)a0 >> k) & 1 ? v : 0.f;
    case 10: {
        if (aug_u01(h0) >= a0) return v;
        const float sn = sinf(1.5707963f * aug_u01(aug_hash(h0, seed ^ 0xDAA66D2Bu)));
        const float beta = sn * sn, dev = fabsf(beta - 0.5f);            // Beta(0.5, 0.5): the arcsine law
        return 255.f * (a2 == 1.f ? 0.5f + dev : (a2 == 2.f ? 0.5f - dev : beta));
    }
    case 11: return 255.f - v;
    case 12: return v >= a0 ? 255.f - v : v;
    default: return v;
    }
}
// RGB <-> CIE L*a*b* on 8-bit values as cv2.COLOR_RGB2Lab / COLOR_Lab2RGB define them for uint8 images (sRGB transfer function, D65 white,
// L * 255 / 100, a + 128, b + 128) - in float arithmetic; OpenCV's own 8-bit path is a fixed-point approximation of the same formulas
// and differs from this by a level here and there (stated in INTEGRATION.md: an approximation, unpinned).
__device__ __forceinline__ float aug_lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.0f / 116.0f; }
__device__ __forceinline__ void aug_rgb2lab(int r, int g, int b, int& L, int& A, int& B) {
    auto lin = [](int c) { const float v = (float)c * (1.0f / 255.f); return v > 0.04045f ? powf((v + 0.055f) * (1.0f / 1.055f), 2.4f) : v * (1.0f / 12.92f); };
    const float fr = lin(r), fg = lin(g), fb = lin(b);
    const float X = (0.412453f * fr + 0.357580f * fg + 0.180423f * fb) * (1.0f / 0.950456f);
    const float Y = 0.212671f * fr + 0.715160f * fg + 0.072169f * fb;
    const float Z = (0.019334f * fr + 0.119193f * fg + 0.950227f * fb) * (1.0f / 1.088754f);
    const float fx = aug_lab_f(X), fy = aug_lab_f(Y), fz = aug_lab_f(Z);
    const float l = Y > 0.008856f ? 116.0f * fy - 16.0f : 903.3f * Y;
    L = aug_clampi((int)rintf(l * 2.55f), 256);
    A = aug_clampi((int)rintf(500.0f * (fx - fy) + 128.0f), 256);
    B = aug_clampi((int)rintf(200.0f * (fy - fz) + 128.0f), 256);
}
__device__ __forceinline__ void aug_lab2rgb(int L, int A, int B, int& r, int& g, int& b) {
    const float l = (float)L * (100.0f / 255.f), a = (float)A - 128.f, bb = (float)B - 128.f;
    float fy, Y;
    if (l <= 8.0f) { Y = l * (1.0f / 903.3f); fy = 7.787f * Y + 16.0f / 116.0f; }
    else { fy = (l + 16.0f) * (1.0f / 116.0f); Y = fy * fy * fy; }
    const float fx = a * (1.0f / 500.0f) + fy, fz = fy - bb * (1.0f / 200.0f);
    auto inv = [](float f) { return f <= 0.2068966f ? (f - 16.0f / 116.0f) * (1.0f / 7.787f) : f * f * f; };
    const float X = inv(fx) * 0.950456f, Z = inv(fz) * 1.088754f;
    const float fr = 3.240479f * X - 1.53715f * Y - 0.498535f * Z;
    const float fg = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
    const float fb = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
    auto gam = [](float v) { v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v); return v > 0.0031308f ? 1.055f * powf(v, 1.0f / 2.4f) - 0.055f : 12.92f * v; };
    r = aug_clampi((int)rintf(gam(fr) * 255.f), 256); g = aug_clampi((int)rintf(gam(fg) * 255.f), 256); b = aug_clampi((int)rintf(gam(fb) * 255.f), 256);
}
constexpr int AUG_CLAHE_MAX_TILES = 12;                 // imgaug draws 3 .. 12 tiles a side
constexpr long AUG_CLAHE_BYTES = (long)AUG_CLAHE_MAX_TILES * AUG_CLAHE_MAX_TILES * 256 + 4 * 256 * 4;      // tile tables + one histogram per wave
// one workgroup per (sample, view): img uint8 [B, H, W, 3], params fp32 [B, 2, AUG_NP] -> staged uint8 [B, 2, H, W, 3]
__global__ __launch_bounds__(256) void augment_spatial_kernel(const unsigned char* __restrict__ img, const float* __restrict__ params,
                                                              unsigned char* __restrict__ staged, int H, int W,
                                                              const unsigned short* __restrict__ overlay, int overlay_layers) {
    char* smem = dynamic_smem();
    const int t = threadIdx.x, b = blockIdx.x >> 1, npix = H * W;
    const float* p = params + (long)blockIdx.x * AUG_NP;
    const long img_bytes = aug_image_bytes(H, W);
    unsigned char* cur = reinterpret_cast<unsigned char*>(smem);
    unsigned char* alt = cur + img_bytes;
    const int Hp = aug_pad16(H), Wp = aug_pad16(W), Hc = Hp / 2, Wc = Wp / 2;
    float* plane = reinterpret_cast<float*>(smem + 2 * img_bytes);                     // Y [Hp][Wp], Cb [Hc][Wc], Cr [Hc][Wc]
    float* coef = plane + Hp * Wp + 2 * Hc * Wc;
    float* dct = coef + Hp * Wp + 2 * Hc * Wc;                                          // D[u][x]
    const unsigned char* src = img + (long)b * npix * 3;
    const unsigned seed = (unsigned)p[AUG_P_SEED];
    const bool pre_invert = p[AUG_P_PREINV] != 0.f;
    for (int i = t; i < npix * 3; i += 256) cur[i] = pre_invert ? (unsigned char)(255 - src[i]) : src[i];
    __syncthreads();
    // a 3 x 3 correlation cur -> alt, then the buffers swap
    auto filter3 = [&](const float* kern, bool pil, float offset, float scale) {
        for (int i = t; i < npix; i += 256) {
            const int y = i / W, x = i % W;
            const bool edge = pil && (y == 0 || x == 0 || y == H - 1 || x == W - 1);
            for (int k = 0; k < 3; ++k) {
                float acc = 0.f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                        acc = fmaf(kern[3 * (dy + 1) + (dx + 1)],
                                   (float)cur[(aug_reflect101(y + dy, H) * W + aug_reflect101(x + dx, W)) * 3 + k], acc);
                alt[i * 3 + k] = edge ? cur[i * 3 + k] : aug_round_u8(pil ? acc / scale + offset : acc);
            }
        }
        __syncthreads();
        unsigned char* tmp = cur; cur = alt; alt = tmp;
    };
    // ---------------------------------------------------------------- group `arithmetic`
    const int opA = (int)p[AUG_P_A];
    if (opA == 13) {
        const int quality = (int)p[AUG_P_A + 1];
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

        __syncthreads();
    } else if (opA == 14 || opA == 15) {
        filter3(p + AUG_P_AK, opA == 15, p[AUG_P_A + 1], p[AUG_P_A + 2] != 0.f ? p[AUG_P_A + 2] : 1.f);
    } else if (opA != 0) {
        for (int i = t; i < npix; i += 256) {
            const int y = i / W, x = i % W;
#pragma unroll
            for (int k = 0; k < 3; ++k) cur[i * 3 + k] = aug_round_u8(aug_arith_point(p, opA, (float)cur[i * 3 + k], y, x, k, H, W, seed));
        }
        __syncthreads();
    }
    // ---------------------------------------------------------------- group `color`
    const int opB = (int)p[AUG_P_B];
    if (opB == 6) {
        // KMeansColorQuantization (imgaug: quantize_colors_kmeans in 8-bit Lab - to_colorspace = [RGB, Lab] -, cv2.kmeans with random
        // centres, <= 10 iterations, eps 1.0, one attempt): Lloyd's iteration on the Lab triples.  k = b0 centres start uniformly inside the
        // data's bounding box widened by a third a side (cv2's generateRandomCenter), an empty cluster takes the point of the most
        // populous cluster that lies farthest from its centre; every pixel becomes its centre, rounded.  Integer sums per cluster: the
        // means do not depend on the order of the additions (the numpy restatement reproduces them exactly).  cv2's own RNG stream is
        // not reproduced (nothing of the reference's augmentation stream is).
        const int k = aug_clampi((int)p[AUG_P_B + 1], 17) < 2 ? 2 : aug_clampi((int)p[AUG_P_B + 1], 17);
        unsigned char* label = reinterpret_cast<unsigned char*>(plane);                  // [npix]
        int* sums = reinterpret_cast<int*>(label + ((npix + 15) / 16) * 16);             // [16][4]: L, a, b sums and the count
        float* cen = reinterpret_cast<float*>(sums + 64);                                // [16][3]
        int* box = reinterpret_cast<int*>(cen + 48);                                     // [3][2] min, max; [6] = changed flag
        float* fard = reinterpret_cast<float*>(box + 8);                                 // [4 waves]: farthest distance / its pixel
        int* fari = reinterpret_cast<int*>(fard + 4);
        for (int i = t; i < npix; i += 256) {
            int L, A, B;
            aug_rgb2lab(cur[3 * i], cur[3 * i + 1], cur[3 * i + 2], L, A, B);
            cur[3 * i] = (unsigned char)L; cur[3 * i + 1] = (unsigned char)A; cur[3 * i + 2] = (unsigned char)B;
        }
        if (t < 3) { box[2 * t] = 255; box[2 * t + 1] = 0; }
        __syncthreads();
        for (int i = t; i < npix * 3; i += 256) { atomicMin(&box[2 * (i % 3)], (int)cur[i]); atomicMax(&box[2 * (i % 3) + 1], (int)cur[i]); }
        __syncthreads();
        if (t < 3 * k) {
            const int j = t / 3, d = t % 3;
            const float u = aug_u01(aug_hash(seed ^ 0x6b6d6e73u, (unsigned)t));
            cen[3 * j + d] = (u * (1.0f + 2.0f / 3.0f) - 1.0f / 3.0f) * (float)(box[2 * d + 1] - box[2 * d]) + (float)box[2 * d];
        }
        __syncthreads();
        for (int iter = 0; iter < 10; ++iter) {
            if (t < 64) sums[t] = 0;
            __syncthreads();
            for (int i = t; i < npix; i += 256) {
                const float x = (float)cur[3 * i], y = (float)cur[3 * i + 1], z = (float)cur[3 * i + 2];
                int best = 0;
                float bd = 3.0e38f;
                for (int j = 0; j < k; ++j) {
                    const float dx = x - cen[3 * j], dy = y - cen[3 * j + 1], dz = z - cen[3 * j + 2];
                    const float dd = (dx * dx + dy * dy) + dz * dz;
                    if (dd < bd) { bd = dd; best = j; }
                }
                label[i] = (unsigned char)best;
                atomicAdd(&sums[4 * best], (int)cur[3 * i]); atomicAdd(&sums[4 * best + 1], (int)cur[3 * i + 1]);
                atomicAdd(&sums[4 * best + 2], (int)cur[3 * i + 2]); atomicAdd(&sums[4 * best + 3], 1);
            }
            __syncthreads();
            // empty clusters, one at a time in index order: the farthest point of the most populous cluster moves over
            for (int j = 0; j < k; ++j) {
                if (sums[4 * j + 3] != 0) continue;                 // (uniform: everybody reads the same counts)
                int big = 0;
                for (int q = 1; q < k; ++q) if (sums[4 * q + 3] > sums[4 * big + 3]) big = q;
                const float c0 = (float)sums[4 * big] / (float)sums[4 * big + 3], c1 = (float)sums[4 * big + 1] / (float)sums[4 * big + 3],
                            c2 = (float)sums[4 * big + 2] / (float)sums[4 * big + 3];
                float fd = -1.f;
                int fi = 0x7fffffff;
                for (int i = t; i < npix; i += 256)
                    if (label[i] == big) {
                        const float dx = (float)cur[3 * i] - c0, dy = (float)cur[3 * i + 1] - c1, dz = (float)cur[3 * i + 2] - c2;
                        const float dd = (dx * dx + dy * dy) + dz * dz;
                        if (dd > fd || (dd == fd && i < fi)) { fd = dd; fi = i; }
                    }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    const float od = shfl_xor(fd, m);
                    const int oi = shfl_xor(fi, m);
                    if (od > fd || (od == fd && oi < fi)) { fd = od; fi = oi; }
                }
                if ((t & 63) == 0) { fard[t >> 6] = fd; fari[t >> 6] = fi; }
                __syncthreads();
                if (t == 0) {
                    float bd2 = fard[0];
                    int bi = fari[0];
                    for (int q = 1; q < 4; ++q) if (fard[q] > bd2 || (fard[q] == bd2 && fari[q] < bi)) { bd2 = fard[q]; bi = fari[q]; }
                    label[bi] = (unsigned char)j;
                    for (int d = 0; d < 3; ++d) { sums[4 * big + d] -= (int)cur[3 * bi + d]; sums[4 * j + d] = (int)cur[3 * bi + d]; }
                    sums[4 * big + 3] -= 1; sums[4 * j + 3] = 1;
                }
                __syncthreads();
            }
            if (t == 0) box[6] = 0;
            __syncthreads();
            if (t < k) {
                float shift = 0.f;
                for (int d = 0; d < 3; ++d) {
                    const float nc = (float)sums[4 * t + d] / (float)sums[4 * t + 3];
                    const float dd = nc - cen[3 * t + d];
                    shift += dd * dd;
                    cen[3 * t + d] = nc;
                }
                if (shift > 1.0f) atomicMax(&box[6], 1);           // eps = 1.0 on the centres' squared movement
            }
            __syncthreads();
            if (box[6] == 0) break;
        }
        for (int i = t; i < npix; i += 256) {                 // (the labels of the FINAL centres, as cv2.kmeans returns them)
            const float x = (float)cur[3 * i], y = (float)cur[3 * i + 1], z = (float)cur[3 * i + 2];
            int j = 0;
            float bd = 3.0e38f;
            for (int q = 0; q < k; ++q) {
                const float dx = x - cen[3 * q], dy = y - cen[3 * q + 1], dz = z - cen[3 * q + 2];
                const float dd = (dx * dx + dy * dy) + dz * dz;
                if (dd < bd) { bd = dd; j = q; }
            }
            int r, g, b;
            aug_lab2rgb(aug_clampi((int)rintf(cen[3 * j]), 256), aug_clampi((int)rintf(cen[3 * j + 1]), 256), aug_clampi((int)rintf(cen[3 * j + 2]), 256), r, g, b);
            cur[3 * i] = (unsigned char)r; cur[3 * i + 1] = (unsigned char)g; cur[3 * i + 2] = (unsigned char)b;
        }
        __syncthreads();
    } else if (opB != 0) {
        const float b0 = p[AUG_P_B + 1], b1 = p[AUG_P_B + 2], b2 = p[AUG_P_B + 3];
        for (int i = t; i < npix; i += 256) {
            int r = cur[i * 3], g = cur[i * 3 + 1], bl = cur[i * 3 + 2];
            if (opB == 1 || opB == 3 || opB == 4) {
                int h, s, v;
                aug_rgb2hsv(r, g, bl, h, s, v);
                if (opB == 1) {
                    h = h + (int)b0; h = h > 255 ? 255 : h;
                } else if (opB == 3) {
                    int h255 = (int)rintf((float)h * (255.f / 180.f));
                    h255 = (int)rintf((float)h255 * b0) % 255;
                    if (h255 < 0) h255 += 255;
                    h = (int)rintf((float)h255 * (180.f / 255.f));
                    s = aug_clampi((int)rintf((float)s * b1), 256);
                } else {
                    h = (h + (int)b0) % 180;
                    if (h < 0) h += 180;
                    s = aug_clampi(s + (int)b1, 256);
                }
                aug_hsv2rgb(h, s, v, r, g, bl);
            } else if (opB == 2) {
                r = aug_round_u8((float)r * b0 + b1); g = aug_round_u8((float)g * b0 + b1); bl = aug_round_u8((float)bl * b0 + b1);
            } else if (opB == 5) {                          // cv2 RGB2GRAY (14-bit fixed point), blended with alpha
                const float gray = (float)((r * 4899 + g * 9617 + bl * 1868 + 8192) >> 14);
                r = aug_round_u8((float)r + b0 * (gray - (float)r)); g = aug_round_u8((float)g + b0 * (gray - (float)g));
                bl = aug_round_u8((float)bl + b0 * (gray - (float)bl));
            } else if (opB == 7) {                          // bin centres of b0 equal bins
                const float q = 256.f / b0;
                r = aug_trunc_u8(floorf((float)r / q) * q + 0.5f * q); g = aug_trunc_u8(floorf((float)g / q) * q + 0.5f * q);
                bl = aug_trunc_u8(floorf((float)bl / q) * q + 0.5f * q);
            } else if (opB == 8) {
                r = aug_round_u8((float)r * b0); g = aug_round_u8((float)g * b1); bl = aug_round_u8((float)bl * b2);
            } else if (opB == 9) {
                const int perm = (int)b0, p0 = perm >> 1, rest0 = p0 == 0 ? 1 : 0, rest1 = p0 == 2 ? 1 : 2;
                const int p1 = (perm & 1) ? rest1 : rest0, p2 = (perm & 1) ? rest0 : rest1;
                const int c[3] = {r, g, bl};
                r = c[p0]; g = c[p1]; bl = c[p2];
            }
            cur[i * 3] = (unsigned char)r; cur[i * 3 + 1] = (unsigned char)g; cur[i * 3 + 2] = (unsigned char)bl;
        }
        __syncthreads();
    }
    // ---------------------------------------------------------------- group `Blur`
    const int mode = (int)p[AUG_P_C];
    if (mode != 0) {
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
                for (int k = 0; k < 3; ++k) alt[i * 3 + k] = aug_round_u8(acc[k]);
            } else if (mode == 2) {                              // median of the k x k window: the value of rank (k k) / 2
                const int r = (int)p[AUG_P_C + 1] / 2, n = (2 * r + 1) * (2 * r + 1), want = n / 2;
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
                    alt[i * 3 + k] = (unsigned char)med;
                }
            } else {                                             // bilateral (OpenCV: L1 colour distance, circular window)
                const int radius = (int)p[AUG_P_C + 1] / 2;
                const float sc_ = p[AUG_P_C + 2], ss_ = p[AUG_P_C + 3];
                const float gc = -0.5f / (sc_ * sc_), gs = -0.5f / (ss_ * ss_);
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
                for (int k = 0; k < 3; ++k) alt[i * 3 + k] = aug_round_u8(num[k] / den);
            }
        }
        __syncthreads();
        unsigned char* tmp = cur; cur = alt; alt = tmp;
    }
    // ---------------------------------------------------------------- group `contrast`
    const int opD = (int)p[AUG_P_D];
    if (opD >= 1 && opD <= 4) {
        const float d0 = p[AUG_P_D + 1], d1 = p[AUG_P_D + 2];
        for (int i = t; i < npix * 3; i += 256) {
            const float v = (float)cur[i], u = v * (1.0f / 255.f);
            float o;
            if (opD == 1) o = 255.f * powf(u, d0);
            else if (opD == 2) o = 127.f + d0 * (v - 127.f);
            else if (opD == 3) o = 255.f / (1.f + expf(d0 * (d1 - u)));
            else o = 255.f * d0 * log2f(1.f + u);
            cur[i] = aug_trunc_u8(o);
        }
        __syncthreads();
    } else if (opD >= 5 && opD <= 8) {
        // 5 HistogramEqualization / 7 CLAHE: on the L channel of cv2's 8-bit Lab; 6 AllChannelsHistogramEqualization / 8 AllChannelsCLAHE:
        // on every channel of the RGB image
        const bool lab = opD == 5 || opD == 7;
        const int nch = lab ? 1 : 3;
        if (lab) {
            for (int i = t; i < npix; i += 256) {
                int L, A, B;
                aug_rgb2lab(cur[3 * i], cur[3 * i + 1], cur[3 * i + 2], L, A, B);
                cur[3 * i] = (unsigned char)L; cur[3 * i + 1] = (unsigned char)A; cur[3 * i + 2] = (unsigned char)B;
            }
            __syncthreads();
        }
        if (opD <= 6) {                                      // cv2.equalizeHist per channel
            int* hist = reinterpret_cast<int*>(plane);           // [3][256]
            unsigned char* lut = reinterpret_cast<unsigned char*>(hist + 768);
            for (int i = t; i < 768; i += 256) hist[i] = 0;
            __syncthreads();
            for (int i = t; i < npix * 3; i += 256)
                if (i % 3 < nch) atomicAdd(&hist[(i % 3) * 256 + cur[i]], 1);
            __syncthreads();
            if (t < nch) {
                const int* hc = hist + 256 * t;
                unsigned char* lc = lut + 256 * t;
                int i0 = 0;
                while (hc[i0] == 0) ++i0;
                if (hc[i0] == npix) {
                    for (int i = 0; i < 256; ++i) lc[i] = (unsigned char)i0;
                } else {
                    const float scale = 255.f / (float)(npix - hc[i0]);
                    int sum = 0;
                    for (int i = 0; i <= i0; ++i) lc[i] = 0;
                    for (int i = i0 + 1; i < 256; ++i) {
                        sum += hc[i];
                        lc[i] = (unsigned char)aug_clampi((int)rintf((float)sum * scale), 256);
                    }
                }
            }
            __syncthreads();
            for (int i = t; i < npix * 3; i += 256)
                if (i % 3 < nch) cur[i] = lut[(i % 3) * 256 + cur[i]];
            __syncthreads();
        } else {
            // cv2.createCLAHE(clipLimit, tileGridSize = (n, n)).apply(channel) (modules/imgproc/src/clahe.cpp): the image padded to whole
            // tiles (BORDER_REFLECT_101), per tile a clipped histogram whose excess is spread over all bins, its cumulative table, and
            // bilinear interpolation between the four nearest tiles' tables.  One wave per tile, four bins per lane.
            const float clip = p[AUG_P_D + 1];
            int tn = (int)p[AUG_P_D + 2];
            tn = tn < 1 ? 1 : (tn > AUG_CLAHE_MAX_TILES ? AUG_CLAHE_MAX_TILES : tn);
            const bool whole = (W % tn == 0) && (H % tn == 0);
            const int We = whole ? W : W + (tn - W % tn), He = whole ? H : H + (tn - H % tn);
            const int tw = We / tn, th = He / tn, area = tw * th;
            int climit = 0;
            if (clip > 0.f) { climit = (int)(clip * (float)area / 256.f); climit = climit < 1 ? 1 : climit; }
            const float lut_scale = 255.f / (float)area;
            unsigned char* luts = reinterpret_cast<unsigned char*>(plane);                       // [tn * tn][256]
            int* whist = reinterpret_cast<int*>(luts + AUG_CLAHE_MAX_TILES * AUG_CLAHE_MAX_TILES * 256) + (t >> 6) * 256;
            const int lane = t & 63, wave = t >> 6;
            for (int c = 0; c < nch; ++c) {
                for (int tile = wave; tile < tn * tn; tile += 4) {
                    const int ty = tile / tn, tx = tile % tn;
                    for (int i = lane; i < 256; i += 64) whist[i] = 0;
                    wave_lds_fence();
                    for (int i = lane; i < area; i += 64) {
                        const int y = aug_reflect101(ty * th + i / tw, H), x = aug_reflect101(tx * tw + i % tw, W);
                        atomicAdd(&whist[cur[(y * W + x) * 3 + c]], 1);
                    }
                    wave_lds_fence();
                    int hv[4], clipped = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hv[j] = whist[4 * lane + j];
                        if (climit > 0 && hv[j] > climit) { clipped += hv[j] - climit; hv[j] = climit; }
                    }
                    if (climit > 0) {
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) clipped += shfl_xor(clipped, m);
                        const int batch = clipped / 256;
                        int residual = clipped - batch * 256;
#pragma unroll
                        for (int j = 0; j < 4; ++j) hv[j] += batch;
                        if (residual != 0) {
                            int step = 256 / residual;
                            step = step < 1 ? 1 : step;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int bin = 4 * lane + j;
                                if (bin % step == 0 && bin / step < residual) hv[j] += 1;
                            }
                        }
                    }
                    // cumulative sums over the 256 bins: four per lane, then a scan over the lanes
                    const int own = hv[0] + hv[1] + hv[2] + hv[3];
                    int incl = own;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const int up = shfl(incl, lane - d < 0 ? lane : lane - d);
                        if (lane >= d) incl += up;
                    }
                    int run = incl - own;
                    unsigned char* lt = luts + tile * 256;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        run += hv[j];
                        lt[4 * lane + j] = (unsigned char)aug_clampi((int)rintf((float)run * lut_scale), 256);
                    }
                    wave_lds_fence();
                }
                __syncthreads();
                const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
                for (int i = t; i < npix; i += 256) {
                    const int y = i / W, x = i % W, v = cur[i * 3 + c];
                    const float tyf = (float)y * inv_th - 0.5f, txf = (float)x * inv_tw - 0.5f;
                    int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
                    const float ya = tyf - (float)ty1, xa = txf - (float)tx1;
                    int ty2 = ty1 + 1, tx2 = tx1 + 1;
                    ty1 = ty1 < 0 ? 0 : ty1; tx1 = tx1 < 0 ? 0 : tx1;
                    ty2 = ty2 > tn - 1 ? tn - 1 : ty2; tx2 = tx2 > tn - 1 ? tn - 1 : tx2;
                    const float r0 = (float)luts[(ty1 * tn + tx1) * 256 + v] * (1.0f - xa) + (float)luts[(ty1 * tn + tx2) * 256 + v] * xa;
                    const float r1 = (float)luts[(ty2 * tn + tx1) * 256 + v] * (1.0f - xa) + (float)luts[(ty2 * tn + tx2) * 256 + v] * xa;
                    alt[i] = (unsigned char)aug_clampi((int)rintf(r0 * (1.0f - ya) + r1 * ya), 256);
                }
                __syncthreads();
                for (int i = t; i < npix; i += 256) cur[i * 3 + c] = alt[i];
                __syncthreads();
            }
        }
        if (lab) {
            for (int i = t; i < npix; i += 256) {
                int r, g, b;
                aug_lab2rgb(cur[3 * i], cur[3 * i + 1], cur[3 * i + 2], r, g, b);
                cur[3 * i] = (unsigned char)r; cur[3 * i + 1] = (unsigned char)g; cur[3 * i + 2] = (unsigned char)b;
            }
            __syncthreads();
        }
    }
    // ---------------------------------------------------------------- group `weather`: cloud layers (Fog / Clouds), one after the other
    const int nlay = (int)p[AUG_P_W], lay0 = (int)p[AUG_P_W + 1], snow = (int)p[AUG_P_W + 2];
    if (overlay && nlay > 0 && lay0 >= 0 && lay0 + nlay <= overlay_layers) {
        for (int l = 0; l < nlay; ++l) {
            const unsigned short* al = overlay + (long)(lay0 + l) * 2 * npix;
            const unsigned short* in = al + npix;
            for (int i = t; i < npix; i += 256) {
                const float a = half2f(al[i]), it = half2f(in[i]);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float c = (float)cur[i * 3 + k];
                    if (snow) {              // SnowflakesLayer: blend by sum (plane 0), then by maximum (plane 1)
                        float v = c + a;
                        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
                        cur[i * 3 + k] = aug_round_u8(v > it ? v : it);
                    } else {                 // CloudLayer / RainLayer: alpha (plane 0) towards an intensity (plane 1)
                        cur[i * 3 + k] = aug_trunc_u8((1.0f - a) * c + a * it);
                    }
                }
            }
            __syncthreads();
        }
    }
    unsigned char* dst = staged + (long)blockIdx.x * npix * 3;
    for (int i = t; i < npix * 3; i += 256) dst[i] = cur[i];
}

// img uint8 [B, H, W, 3]; staged uint8 [B, 2, H, W, 3] (augment_spatial_kernel: the augmented images of views 1 / 2);
// theta fp32 [B, 3, 3]; out fp32 [B, 3 views, 3, H, W]: view 0 plain, view 1 augmented, view 2 augmented and warped (bilinear, zero fill,
// src = theta (x, y, 1) in normalised coordinates - the inverse of how the dataset derives theta, datasetsupervised_kmeans.py:65-71)
// warp_maps (optional) fp32 [maps, 2, H, W]: per output pixel the SOURCE position (x, y) in pixels - the piecewise-affine member of the
// finetuning geometry (imgaug PiecewiseAffine: a jittered 4 x 4 mesh, one affine map per Delaunay triangle, drawn on the host:
// ccd_amd/dataset/weather.py).  A sample whose view-2 parameter row has params[AUG_P_W + 3] = m > 0 samples map m - 1 instead of theta.
__global__ __launch_bounds__(256) void augment_views_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ staged,
                                                            const float* __restrict__ theta, float* __restrict__ out,
                                                            int B, int H, int W, float m0, float m1, float m2, float is0,
                                                            float is1, float is2, const float* __restrict__ params,
                                                            const float* __restrict__ warp_maps, int warp_count) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int y = pix / W, x = pix % W;
    const unsigned char* src = img + (long)b * H * W * 3;
    const float mean[3] = {m0, m1, m2}, istd[3] = {is0, is1, is2};
    float* o = out + (long)b * 9 * H * W + pix;
    const long plane = (long)H * W;
    const unsigned char* s1 = staged + (long)(2 * b) * H * W * 3 + pix * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k * plane] = ((float)src[pix * 3 + k] * (1.0f / 255.f) - mean[k]) * istd[k];
        o[(3 + k) * plane] = ((float)s1[k] * (1.0f / 255.f) - mean[k]) * istd[k];
    }
    const float* th = theta + (long)b * 9;
    const unsigned char* s2 = staged + (long)(2 * b + 1) * H * W * 3;
    const float xn = 2.0f * (float)x / (float)(W - 1) - 1.0f, yn = 2.0f * (float)y / (float)(H - 1) - 1.0f;
    float xs = ((th[0] * xn + th[1] * yn + th[2]) + 1.0f) * 0.5f * (float)(W - 1);
    float ys = ((th[3] * xn + th[4] * yn + th[5]) + 1.0f) * 0.5f * (float)(H - 1);
    if (warp_maps) {
        const int m = (int)params[((long)b * 2 + 1) * AUG_NP + AUG_P_W + 3];
        if (m > 0 && m <= warp_count) {
            const float* wm = warp_maps + (long)(m - 1) * 2 * plane;
            xs = wm[pix];
            ys = wm[plane + pix];
        }
    }
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
                const int sp = (yy * W + xx) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k] += wgt * (float)s2[sp + k];
            }
        }
#pragma unroll
    for (int k = 0; k < 3; ++k) o[(6 + k) * plane] = (acc[k] * (1.0f / 255.f) - mean[k]) * istd[k];
}

}  // namespace ccd
