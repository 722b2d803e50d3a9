"""CPU restatement (numpy) of the data-pipeline steps - TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline);
the product never imports it.

* kmeans2_mask: `clusterpixels(im, 2)` of the reference, mask_create/generate_mask.py:13-29 (same function in
  Dino/utils/kmeans.py:7-23): scipy.cluster.vq.kmeans (20 random restarts of Lloyd, best kept) + vq + the border rule.
  Restated deterministically: on one axis a clustering is a threshold; Lloyd's iteration can stop at any threshold that
  is a fixed point (every present value on its own side of the centroids' midpoint), and scipy keeps the restart with the
  smallest MEAN distance (vq distances are not squared) - so: all fixed points enumerated, smallest mean absolute distance
  kept, code 1 = brighter cluster, flipped when >= 3 of the 4 border lines are mostly 1.  PINNED: bit-equal to the real
  function on the 48 word images of tests/golden/kmeans_masks.npz (tools/gen_golden.py `gen_kmeans`, three numpy seeds per
  image agree); the squared-error optimum (Otsu) differs on 10 of them.  Where the border rule is undecided for BOTH orientations (num < 3 either way)
  the reference's output depends on its random initial centroids; this restatement keeps "brighter = 1".
* gray_from_rgb: PIL's `convert("L")` (ITU-R 601-2 luma in 16.16 fixed point), generate_mask.py:69.
* augment_views: ccd_amd/csrc/kernels/datapipe.h::augment_views_kernel in float32 (the reference's imgaug chain is a
  random process with its own generators - there is no value parity to pin, only the tensor contract and the theta
  geometry of datasetsupervised_kmeans.py:65-71, which tests/test_datapipe_cpu.py checks against the reference formula).
"""
import numpy as np


def gray_from_rgb(rgb):
    rgb = np.asarray(rgb, dtype=np.uint32)
    return ((rgb[..., 0] * 19595 + rgb[..., 1] * 38470 + rgb[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def kmeans2_mask(gray):
    gray = np.asarray(gray, dtype=np.uint8)
    h, w = gray.shape
    hist = [int(c) for c in np.bincount(gray.reshape(-1), minlength=256)]
    ntot = sum(hist)
    stot = sum(c * v for v, c in enumerate(hist))
    best = None
    n0 = s0 = 0
    for g in range(256):
        n0 += hist[g]
        s0 += hist[g] * g
        n1, s1 = ntot - n0, stot - s0
        if hist[g] == 0 or n1 == 0:
            continue
        nxt = next(v for v in range(g + 1, 256) if hist[v] > 0)
        m0, m1 = float(s0) / float(n0), float(s1) / float(n1)
        mid = (m0 + m1) / 2.0
        if not (float(g) < mid < float(nxt)):          # not a fixed point of Lloyd's iteration
            continue
        acc = 0.0
        for v in range(256):                             # sequential fp64 sum, the kernel's order
            d = float(v) - (m0 if v <= g else m1)
            acc += float(hist[v]) * (-d if d < 0.0 else d)
        mad = acc / float(ntot)
        if best is None or mad < best[0]:
            best = (mad, g)
    if best is None:
        return np.zeros((h, w), np.uint8)
    g = best[1]
    code = (gray > g).astype(np.int64)
    fc, lc, fr, lr = code[:, 0].sum(), code[:, -1].sum(), code[0, :].sum(), code[-1, :].sum()
    num = int(fr > w // 2) + int(lr > w // 2) + int(fc > h // 2) + int(lc > h // 2)
    return (1 - code if num >= 3 else code).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------- augmentation
AUG_NP = 96


def _hash(a, b):
    a = np.asarray(a, dtype=np.uint64) & 0xFFFFFFFF
    b = np.asarray(b, dtype=np.uint64) & 0xFFFFFFFF
    z = ((a * 0x9E3779B1) & 0xFFFFFFFF) ^ ((b + 0x7F4A7C15) & 0xFFFFFFFF)
    z ^= z >> 16; z = (z * 0x85EBCA6B) & 0xFFFFFFFF; z ^= z >> 13; z = (z * 0xC2B2AE35) & 0xFFFFFFFF; z ^= z >> 16
    return z


def _u01(h):
    return (h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)


def colour(p, rgb, pix_id):
    """p [16] fp32, rgb [...,3] fp32 in 0..255, pix_id [...] -> [...,3]."""
    f = np.float32
    c = rgb.astype(f).copy()
    if p[0] != 0:
        c = f(255) - c
    if p[12] < 256:
        c = np.where(c >= p[12], f(255) - c, c)
    gray = f(0.299) * c[..., 0] + f(0.587) * c[..., 1] + f(0.114) * c[..., 2]
    c = c + f(p[1]) * (gray[..., None] - c)
    perm = int(p[2])
    p0 = perm >> 1
    rest = [k for k in range(3) if k != p0]
    order = [p0, rest[1], rest[0]] if perm & 1 else [p0, rest[0], rest[1]]
    s = c[..., order]
    seed = int(p[13]) & 0xFFFFFFFF
    out = np.empty_like(s)
    for k in range(3):
        v = s[..., k]
        if p[3] != 1.0:
            v = f(255) * np.power(np.maximum(v, f(0)) * f(1.0 / 255.0), f(p[3])).astype(f)
        v = v * f(p[4 + k])
        v = f(128) + f(p[7]) * (v - f(128)) + f(p[8])
        h0 = _hash(np.asarray(pix_id, dtype=np.uint64) * 3 + k, seed)
        if p[9] > 0:
            h1 = _hash(h0, seed ^ 0xA511E9B3)
            v = v + f(p[9]) * np.sqrt(f(-2) * np.log(np.maximum(_u01(h0), f(1e-7)))) * np.cos(f(6.2831853) * _u01(h1))
        if p[10] > 0:
            v = v * (f(1) + f(p[10]) * (f(2) * _u01(_hash(h0, seed ^ 0x3C6EF372)) - f(1)))
        if p[11] > 0:
            u = _u01(_hash(h0, seed ^ 0xDAA66D2B))
            v = np.where(u < f(p[11]), np.where(u < f(0.5) * f(p[11]), f(0), f(255)), v)
        out[..., k] = np.clip(v, 0, 255)
    return out.astype(f)


def staged_source(p, src_u8):
    """augment_spatial_kernel: the neighbourhood members of one (sample, view): JPEG round trip first (p[25] = quality), then
    the Blur-group member p[14] selects - 1: 7 x 7 correlation p[32:81], 2: median k = p[15], 3: bilateral (d = p[15],
    sigma_color = p[26], sigma_space = p[27]).  uint8 [H,W,3] -> uint8 [H,W,3]."""
    cur = np.asarray(src_u8, dtype=np.uint8)
    if int(p[25]) > 0:
        cur = jpeg_roundtrip(cur, int(p[25]))
    mode = int(p[14])
    if mode == 1:
        cur = filter7(cur, np.asarray(p[32:81], dtype=np.float32).reshape(7, 7))
    elif mode == 2:
        cur = median_blur(cur, int(p[15]))
    elif mode == 3:
        cur = bilateral_blur(cur, int(p[15]), float(p[26]), float(p[27]))
    return cur


def augment_views(img, params, theta, mean, std):
    """img uint8 [B,H,W,3], params [B,2,96], theta [B,3,3] -> fp32 [B,3,3,H,W]."""
    f = np.float32
    img = np.asarray(img)
    B, H, W, _ = img.shape
    mean, istd = np.asarray(mean, f), (f(1) / np.asarray(std, f)).astype(f)
    out = np.zeros((B, 3, 3, H, W), f)
    ys, xs = np.mgrid[0:H, 0:W]
    pid = (ys * W + xs).astype(np.uint64)

    def norm(c):          # [H,W,3] -> [3,H,W]
        return ((c * f(1.0 / 255.0) - mean) * istd).astype(f).transpose(2, 0, 1)

    for b in range(B):
        src = img[b].astype(f)
        out[b, 0] = norm(src)
        out[b, 1] = norm(colour(params[b, 0], staged_source(params[b, 0], img[b]).astype(f), pid))
        th = theta[b].astype(f)
        xn = f(2) * xs.astype(f) / f(W - 1) - f(1)
        yn = f(2) * ys.astype(f) / f(H - 1) - f(1)
        sx = ((th[0, 0] * xn + th[0, 1] * yn + th[0, 2]) + f(1)) * f(0.5) * f(W - 1)
        sy = ((th[1, 0] * xn + th[1, 1] * yn + th[1, 2]) + f(1)) * f(0.5) * f(H - 1)
        x0, y0 = np.floor(sx), np.floor(sy)
        ax, ay = (sx - x0).astype(f), (sy - y0).astype(f)
        x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
        col2 = colour(params[b, 1], staged_source(params[b, 1], img[b]).astype(f), pid)   # colour of every source pixel once
        acc = np.zeros((H, W, 3), f)
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy = x0 + dx, y0 + dy
                wgt = (ax if dx else f(1) - ax) * (ay if dy else f(1) - ay)
                ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (wgt != 0)
                tap = col2[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
                acc = acc + np.where(ok[..., None], wgt[..., None] * tap, f(0))
        out[b, 2] = norm(acc)
    return out


# ------------------------------------------------------------------------------------- spatial members of the pipelines
# Restatements of the DOCUMENTED algorithms behind the imgaug members that need a neighbourhood (augmentation_pipelines.py:
# Blur group :165-176, JpegCompression :140): imgaug itself is absent here, so each function is pinned against the library
# that imgaug delegates to where that library exists in this image (PIL / libjpeg for JPEG, scipy.ndimage for the median and
# the correlation) - tests/test_datapipe_cpu.py - and restates OpenCV's documented behaviour where it does not (bilateral).

def reflect101(idx, n):
    """cv2.BORDER_REFLECT_101 (the default border of filter2D / blur / bilateralFilter): ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ..."""
    idx = np.abs(np.asarray(idx))
    period = 2 * (n - 1) if n > 1 else 1
    idx = idx % period
    return np.where(idx > n - 1, period - idx, idx)


def filter7(img, kern):
    """cv2.filter2D (correlation) with a 7 x 7 kernel [dy + 3][dx + 3], BORDER_REFLECT_101, rounded to uint8 (imgaug's
    convolutional augmenters, GaussianBlur / AverageBlur / MotionBlur kernels embedded in the 7 x 7 grid)."""
    img = np.asarray(img, dtype=np.float32)
    H, W, _ = img.shape
    out = np.zeros_like(img)
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            wgt = np.float32(kern[dy + 3][dx + 3])
            if wgt == 0:
                continue
            out = out + wgt * img[reflect101(np.arange(H) + dy, H)][:, reflect101(np.arange(W) + dx, W)]
    return np.clip(np.floor(out + np.float32(0.5)), 0, 255).astype(np.uint8)


def median_blur(img, k):
    """cv2.medianBlur(img, k) (iaa.MedianBlur): per channel median of the k x k window, BORDER_REPLICATE."""
    img = np.asarray(img, dtype=np.uint8)
    H, W, C = img.shape
    r = k // 2
    stack = []
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            stack.append(img[np.clip(np.arange(H) + dy, 0, H - 1)][:, np.clip(np.arange(W) + dx, 0, W - 1)])
    return np.sort(np.stack(stack, 0), axis=0)[(k * k) // 2].astype(np.uint8)


def bilateral_blur(img, d, sigma_color, sigma_space):
    """cv2.bilateralFilter(img, d, sigmaColor, sigmaSpace) for 8-bit 3-channel images as OpenCV documents / implements it
    (iaa.BilateralBlur): radius = d / 2, taps inside the circle r <= radius, weight = exp(-r^2 / (2 sigma_space^2)) *
    exp(-(|db| + |dg| + |dr|)^2 / (2 sigma_color^2)) - ONE weight for the three channels from the L1 colour distance -
    BORDER_REFLECT_101, result rounded."""
    img = np.asarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    f = np.float32
    radius = int(d) // 2
    gc = f(-0.5) / (f(sigma_color) * f(sigma_color))
    gs = f(-0.5) / (f(sigma_space) * f(sigma_space))
    src = img.astype(f)
    num = np.zeros((H, W, 3), f)
    den = np.zeros((H, W), f)
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            rr = dy * dy + dx * dx
            if rr > radius * radius:
                continue
            tap = src[reflect101(np.arange(H) + dy, H)][:, reflect101(np.arange(W) + dx, W)]
            dist = np.abs(tap - src).sum(-1)
            w = (np.exp(f(rr) * gs) * np.exp(dist * dist * gc)).astype(f)
            num += w[..., None] * tap
            den += w
    return np.clip(np.floor(num / den[..., None] + f(0.5)), 0, 255).astype(np.uint8)


def motion_blur_kernel(k, angle_deg, direction):
    """iaa.MotionBlur(k, angle, direction, order=1): a k x k matrix whose middle COLUMN holds linspace(d, 1 - d) with
    d = (clip(direction, -1, 1) + 1) / 2, quantised to uint8, rotated by `angle` about the matrix centre with bilinear
    sampling and zero padding (iaa.Affine(rotate=angle, order=1) on the uint8 matrix), divided by its sum."""
    k = k if k % 2 else k + 1
    d = (min(max(float(direction), -1.0), 1.0) + 1.0) / 2.0
    m = np.zeros((k, k), np.float64)
    m[:, k // 2] = np.linspace(d, 1.0 - d, num=k)
    m8 = np.floor(m * 255.0).astype(np.float64)                   # (matrix * 255).astype(np.uint8) truncates
    c = (k - 1) / 2.0
    a = np.deg2rad(angle_deg)
    out = np.zeros((k, k), np.float64)
    for y in range(k):
        for x in range(k):
            # inverse map of a rotation by +angle (image coordinates, y down) about the centre
            xs = np.cos(a) * (x - c) + np.sin(a) * (y - c) + c
            ys = -np.sin(a) * (x - c) + np.cos(a) * (y - c) + c
            x0, y0 = int(np.floor(xs)), int(np.floor(ys))
            ax, ay = xs - x0, ys - y0
            v = 0.0
            for yy, wy in ((y0, 1 - ay), (y0 + 1, ay)):
                for xx, wx in ((x0, 1 - ax), (x0 + 1, ax)):
                    if 0 <= yy < k and 0 <= xx < k:
                        v += wy * wx * m8[yy, xx]
            out[y, x] = np.floor(v + 0.5)
    s = out.sum()
    return (out / s if s > 0 else m / m.sum()).astype(np.float32)


JPEG_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29,
                      51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121,
                      120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64).reshape(8, 8)
JPEG_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99,
                        99, 99, 99, 99] + [99] * 32, dtype=np.int64).reshape(8, 8)


def jpeg_quant_table(base, quality):
    """libjpeg's jpeg_quality_scaling + jpeg_add_quant_table (baseline: entries clamped to 1..255)."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    return np.clip((np.asarray(base, np.int64) * scale + 50) // 100, 1, 255)


_DCT = np.array([[(np.sqrt(0.125) if u == 0 else 0.5) * np.cos((2 * x + 1) * u * np.pi / 16.0) for x in range(8)]
                 for u in range(8)], dtype=np.float64)          # F = D f D^T


def jpeg_roundtrip(rgb, quality):
    """PIL's Image.save(format='JPEG', quality=q) -> Image.open() as baseline JFIF does it (iaa.JpegCompression goes through
    exactly that): RGB -> YCbCr (libjpeg's 16-bit fixed point), 4:2:0 chroma (h2v2 averages with the alternating 1 / 2 bias,
    edges replicated up to whole 16 x 16 MCUs), 8 x 8 DCT of the level-shifted samples, quantisation by the quality-scaled
    Annex-K tables (round half away from zero), dequantisation, inverse DCT, clamp, "fancy" triangle up-sampling of the chroma
    planes, YCbCr -> RGB.  The DCTs are floating point here; libjpeg's default is the 13-bit integer "islow" pair, which
    lands within one level of it except where a coefficient sits on a rounding boundary."""
    rgb = np.asarray(rgb, dtype=np.int64)
    H, W, _ = rgb.shape
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    pad = rgb[np.minimum(np.arange(Hp), H - 1)][:, np.minimum(np.arange(Wp), W - 1)]
    R, G, B = pad[..., 0], pad[..., 1], pad[..., 2]
    fix = lambda x: int(x * 65536 + 0.5)
    half = 1 << 15
    Y = (fix(0.29900) * R + fix(0.58700) * G + fix(0.11400) * B + half) >> 16
    cbias = (128 << 16) + half - 1
    Cb = (-fix(0.16874) * R - fix(0.33126) * G + fix(0.50000) * B + cbias) >> 16
    Cr = (fix(0.50000) * R - fix(0.41869) * G - fix(0.08131) * B + cbias) >> 16

    def down(P):           # h2v2_downsample: bias 1, 2, 1, 2 ... along a row
        s = P[0::2, 0::2] + P[0::2, 1::2] + P[1::2, 0::2] + P[1::2, 1::2]
        bias = np.where(np.arange(s.shape[1]) % 2 == 0, 1, 2)[None, :]
        return (s + bias) >> 2

    def codec(P, table):
        h, w = P.shape
        blocks = (P.astype(np.float64) - 128.0).reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)
        F = np.einsum("ux,abxy,vy->abuv", _DCT, blocks, _DCT)
        t = table.astype(np.float64)
        q = np.sign(F) * np.floor(np.abs(F) / t + 0.5)             # round half away from zero (jcdctmgr's DIVIDE_BY)
        rec = np.einsum("ux,abuv,vy->abxy", _DCT, q * t, _DCT)
        pix = np.clip(np.floor(rec + 128.0 + 0.5), 0, 255).astype(np.int64)
        return pix.transpose(0, 2, 1, 3).reshape(h, w)

    Yd = codec(Y, jpeg_quant_table(JPEG_LUMA, quality))
    tc = jpeg_quant_table(JPEG_CHROMA, quality)
    Cbd, Crd = codec(down(Cb), tc), codec(down(Cr), tc)

    def up(P):             # h2v2_fancy_upsample (jdsample.c): 3/4 nearer + 1/4 further in each direction, edges replicated
        h, w = P.shape
        above, below = P[np.maximum(np.arange(h) - 1, 0)], P[np.minimum(np.arange(h) + 1, h - 1)]
        out = np.zeros((2 * h, 2 * w), np.int64)
        for v, far in ((0, above), (1, below)):
            col = 3 * P + far                                        # "thiscolsum"
            last = col[:, np.maximum(np.arange(w) - 1, 0)]
            nxt = col[:, np.minimum(np.arange(w) + 1, w - 1)]
            out[v::2, 0::2] = (3 * col + last + 8) >> 4
            out[v::2, 1::2] = (3 * col + nxt + 7) >> 4
        return out

    Cbu, Cru = up(Cbd) - 128, up(Crd) - 128
    r = Yd + ((fix(1.40200) * Cru + half) >> 16)
    g = Yd + ((-fix(0.34414) * Cbu - fix(0.71414) * Cru + half) >> 16)
    b = Yd + ((fix(1.77200) * Cbu + half) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255)[:H, :W].astype(np.uint8)


def jpeg_quality_from_compression(compression):
    """iaa.JpegCompression: compression 0 .. 100 -> PIL quality 100 .. 1 (linear, rounded, clipped)."""
    return int(np.clip(np.round(1 + (100 - 1) * (1.0 - compression / 100.0)), 1, 100))
