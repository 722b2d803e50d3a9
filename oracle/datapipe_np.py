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
* augment_views / staged_source: the reference's imgaug chain (augmentation_pipelines.py:120-205) member by member - each member
  restates the documented behaviour of the library call imgaug makes (cv2 / PIL / numpy), pinned against that library where
  this image has it (PIL: JPEG, the pillike filters; scipy.ndimage: median, correlation) and stated as unpinned where it does
  not (cv2's HSV conversion, equalizeHist, bilateralFilter; imgaug's own tables).  The random members draw from the device
  kernel's counter-based generator, not numpy's: there is no value parity with imgaug's draws, only the distributions and the
  theta geometry of datasetsupervised_kmeans.py:65-71 (tests/test_datapipe_cpu.py).
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


# parameter layout of one (sample, view) row: ccd_amd/csrc/kernels/datapipe.h
P_SEED, P_PREINV, P_A, P_AK, P_B, P_C, P_D, P_KERN, P_W = 0, 1, 2, 9, 18, 24, 28, 32, 81


def _round_u8(v):
    return np.clip(np.floor(np.asarray(v, np.float32) + np.float32(0.5)), 0, 255).astype(np.uint8)


def _trunc_u8(v):
    return np.clip(np.asarray(v, np.float32), 0, 255).astype(np.uint8)


def arith_pointwise(p, img):
    """The pointwise members of the `arithmetic` group (augmentation_pipelines.py:122-144) on a uint8 image [H,W,3]; the random
    members draw from the kernel's counter-based generator, so kernel and restatement agree value by value."""
    f = np.float32
    op, seed = int(p[P_A]), int(p[P_SEED]) & 0xFFFFFFFF
    a0, a1, a2, a3 = (f(p[P_A + 1 + i]) for i in range(4))
    H, W, _ = img.shape
    ys, xs = np.mgrid[0:H, 0:W]
    out = np.empty((H, W, 3), f)
    for k in range(3):
        v = img[..., k].astype(f)
        ch = k if a1 != 0 else 0
        e = ((ys * W + xs) * 3 + ch).astype(np.uint64)
        h0 = _hash(e, seed ^ 0x51ED270B)
        if op == 1:
            v = v + (np.floor(_u01(h0) * (f(2) * a0 + f(1))).astype(np.int64) - int(a0)).astype(f)
        elif op == 2:
            h1 = _hash(h0, seed ^ 0xA511E9B3)
            v = v + a0 * np.sqrt(f(-2) * np.log(np.maximum(_u01(h0), f(1e-7)))) * np.cos(f(6.2831853) * _u01(h1))
        elif op == 3:
            u = _u01(h0) - f(0.5)
            m = np.maximum(f(1) - f(2) * np.abs(u), f(1e-7))
            v = v - a0 * np.where(u < 0, f(-1), f(1)) * np.log(m)
        elif op == 4:
            lim = np.exp(-a0).astype(f)
            prod = np.ones((H, W), f)
            n = np.full((H, W), -1, np.int64)
            h = h0.copy()
            live = np.ones((H, W), bool)
            for _ in range(201):
                prod = np.where(live, (prod * np.maximum(_u01(h), f(1e-7))).astype(f), prod)
                h = np.where(live, _hash(h, seed ^ 0x3C6EF372), h)
                n = np.where(live, n + 1, n)
                live = live & (prod > lim) & (n < 200)
                if not live.any():
                    break
            v = v + n.astype(f)
        elif op == 5:
            v = v * f(p[P_A + 1 + k])
        elif op == 6:
            v = v * (a0 + (a2 - a0) * _u01(h0))
        elif op == 7:
            v = np.where(_u01(h0) < a0, f(0), v)
        elif op == 8:
            rows, cols = int(a2), int(a3)
            cy, cx = np.clip(ys * rows // H, 0, rows - 1), np.clip(xs * cols // W, 0, cols - 1)
            v = np.where(_u01(_hash(((cy * cols + cx) * 3 + ch).astype(np.uint64), seed ^ 0x6A09E667)) < a0, f(0), v)
        elif op == 9:
            v = v if (int(a0) >> k) & 1 else np.zeros_like(v)
        elif op == 10:
            sn = np.sin(f(1.5707963) * _u01(_hash(h0, seed ^ 0xDAA66D2B))).astype(f)
            beta = sn * sn
            dev = np.abs(beta - f(0.5))
            rep = f(255) * (f(0.5) + dev if a2 == 1 else (f(0.5) - dev if a2 == 2 else beta))
            v = np.where(_u01(h0) < a0, rep, v)
        elif op == 11:
            v = f(255) - v
        elif op == 12:
            v = np.where(v >= a0, f(255) - v, v)
        out[..., k] = v
    return _round_u8(out)


def filter3(img, kern, pil=False, offset=0.0, scale=1.0):
    """A 3 x 3 correlation.  pil=False: cv2.filter2D with BORDER_REFLECT_101 (imgaug's Convolve members); pil=True:
    PIL.ImageFilter.Kernel((3, 3), kern, scale, offset) - sum / scale + offset, and the one-pixel border is copied."""
    src = np.asarray(img, dtype=np.float32)
    H, W, _ = src.shape
    kern = np.asarray(kern, np.float32).reshape(3, 3)
    acc = np.zeros_like(src)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            acc = acc + kern[dy + 1, dx + 1] * src[reflect101(np.arange(H) + dy, H)][:, reflect101(np.arange(W) + dx, W)]
    if not pil:
        return _round_u8(acc)
    out = _round_u8(acc / np.float32(scale) + np.float32(offset))
    out[0], out[-1], out[:, 0], out[:, -1] = img[0], img[-1], img[:, 0], img[:, -1]
    return out


def rgb_to_hsv_cv(img):
    """cv2.cvtColor(img, COLOR_RGB2HSV) for uint8 (color_hsv.cpp RGB2HSV_b): v = max, s = diff * sdiv[v], h from the 60-degree
    sector formula * hdiv[diff], 12-bit fixed point with rounded division tables, H in 0..179."""
    c = np.asarray(img, np.int64)
    r, g, b = c[..., 0], c[..., 1], c[..., 2]
    v = c.max(-1)
    diff = v - c.min(-1)
    sdiv = np.where(v > 0, np.rint(1044480.0 / np.maximum(v, 1)), 0).astype(np.int64)
    hdiv = np.where(diff > 0, np.rint(122880.0 / np.maximum(diff, 1)), 0).astype(np.int64)
    s = (diff * sdiv + 2048) >> 12
    hh = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (hh * hdiv + 2048) >> 12
    h = np.where(h < 0, h + 180, h)
    return np.stack([np.where(diff > 0, h, 0), s, v], -1)


def hsv_to_rgb_cv(hsv):
    """cv2.cvtColor(hsv, COLOR_HSV2RGB) for uint8 (HSV2RGB_b -> HSV2RGB_f): float32, h * 6 / 180 reduced into [0, 6), the sector
    table, saturate_cast<uchar>(x * 255)."""
    f = np.float32
    hsv = np.asarray(hsv, np.int64)
    hi, si, vi = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    s, v = si.astype(f) * f(1.0 / 255.0), vi.astype(f) * f(1.0 / 255.0)
    h = hi.astype(f) * f(6.0 / 180.0)
    for _ in range(4):
        h = np.where(h >= f(6), h - f(6), h)
        h = np.where(h < f(0), h + f(6), h)
    sector = np.floor(h).astype(np.int64)
    h = h - sector.astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector, h = np.where(bad, 0, sector), np.where(bad, f(0), h)
    tab = np.stack([v, v * (f(1) - s), v * (f(1) - s * h), v * (f(1) - s * (f(1) - h))], -1).astype(f)
    sector_bgr = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    pick = sector_bgr[sector]
    bgr = np.take_along_axis(tab, pick, -1)
    bgr = np.where((si == 0)[..., None], v[..., None], bgr)
    out = np.clip(np.rint(bgr * f(255)), 0, 255).astype(np.uint8)
    return out[..., ::-1]


def colour_member(p, img):
    """One member of the `color` group (augmentation_pipelines.py:146-163) on a uint8 image."""
    f = np.float32
    op = int(p[P_B])
    b0, b1, b2 = f(p[P_B + 1]), f(p[P_B + 2]), f(p[P_B + 3])
    img = np.asarray(img, np.uint8)
    if op in (1, 3, 4):
        hsv = rgb_to_hsv_cv(img)
        h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
        if op == 1:
            h = np.minimum(h + int(b0), 255)
        elif op == 3:
            h255 = np.rint(h.astype(f) * f(255.0 / 180.0)).astype(np.int64)
            h255 = np.mod(np.rint(h255.astype(f) * b0).astype(np.int64), 255)
            h = np.rint(h255.astype(f) * f(180.0 / 255.0)).astype(np.int64)
            s = np.clip(np.rint(s.astype(f) * b1).astype(np.int64), 0, 255)
        else:
            h = np.mod(h + int(b0), 180)
            s = np.clip(s + int(b1), 0, 255)
        return hsv_to_rgb_cv(np.stack([h, s, v], -1))
    if op == 6:
        return kmeans_quantize(img, int(b0), int(p[P_SEED]))
    c = img.astype(f)
    if op == 2:
        return _round_u8(c * b0 + b1)
    if op == 5:
        i = img.astype(np.int64)
        gray = ((i[..., 0] * 4899 + i[..., 1] * 9617 + i[..., 2] * 1868 + 8192) >> 14).astype(f)
        return _round_u8(c + b0 * (gray[..., None] - c))
    if op == 7:
        q = f(256) / b0
        return _trunc_u8(np.floor(c / q) * q + f(0.5) * q)
    if op == 8:
        return _round_u8(c * np.array([b0, b1, b2], f))
    if op == 9:
        perm = int(b0)
        p0 = perm >> 1
        rest = [k for k in range(3) if k != p0]
        order = [p0, rest[1], rest[0]] if perm & 1 else [p0, rest[0], rest[1]]
        return img[..., order]
    return img


def kmeans_quantize(img, k, seed):
    """KMeansColorQuantization (imgaug.augmenters.color.quantize_colors_kmeans in 8-bit Lab; cv2.kmeans: random centres inside the data's
    bounding box widened by 1/3 a side, <= 10 Lloyd iterations, stop when no centre moves by more than eps = 1 (squared), an empty cluster
    takes the farthest point of the most populous one; labels of the final centres) - the device's counter-based draws for the centres."""
    f = np.float32
    k = min(max(int(k), 2), 16)
    lab = rgb_to_lab_u8(img)
    pts = lab.reshape(-1, 3).astype(np.int64)
    x = pts.astype(f)
    lo, hi = pts.min(0), pts.max(0)
    cen = np.zeros((k, 3), f)
    for t in range(3 * k):
        j, d = t // 3, t % 3
        u = _u01(_hash(np.uint32(seed) ^ np.uint32(0x6b6d6e73), np.uint32(t)))
        cen[j, d] = (f(u) * (f(1.0) + f(2.0) / f(3.0)) - f(1.0) / f(3.0)) * f(hi[d] - lo[d]) + f(lo[d])

    def assign(c):
        d = x[:, None, :] - c[None, :, :]
        dd = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        return np.argmin(dd, axis=1)

    for _ in range(10):
        label = assign(cen)
        sums = np.zeros((k, 3), np.int64)
        cnt = np.zeros(k, np.int64)
        np.add.at(sums, label, pts)
        np.add.at(cnt, label, 1)
        for j in range(k):
            if cnt[j] != 0:
                continue
            big = int(np.argmax(cnt))                                  # (first of the most populous ones)
            c = (sums[big].astype(f) / f(cnt[big])).astype(f)
            idx = np.nonzero(label == big)[0]
            d = x[idx] - c[None]
            dd = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            bi = idx[int(np.argmax(dd))]                               # (the first of the farthest ones)
            label[bi] = j
            sums[big] -= pts[bi]; cnt[big] -= 1
            sums[j] = pts[bi]; cnt[j] = 1
        new = (sums.astype(f) / cnt.astype(f)[:, None]).astype(f)
        dlt = new - cen
        shift = (dlt[:, 0] * dlt[:, 0] + dlt[:, 1] * dlt[:, 1]) + dlt[:, 2] * dlt[:, 2]
        cen = new
        if not (shift > f(1.0)).any():
            break
    label = assign(cen)
    q = np.clip(np.rint(cen), 0, 255).astype(np.uint8)
    return lab_to_rgb_u8(q[label].reshape(lab.shape))


def equalize_hist_cv(channel):
    """cv2.equalizeHist on one uint8 channel: the first occupied bin maps to 0, lut[i] = round(cumsum beyond it * 255 / (total - its count))."""
    ch = np.asarray(channel, np.uint8)
    hist = np.bincount(ch.reshape(-1), minlength=256)
    i0 = int(np.nonzero(hist)[0][0])
    if hist[i0] == ch.size:
        return np.full_like(ch, i0)
    scale = np.float32(255.0) / np.float32(ch.size - hist[i0])
    csum = np.cumsum(np.where(np.arange(256) > i0, hist, 0))
    lut = np.clip(np.rint(csum.astype(np.float32) * scale), 0, 255).astype(np.uint8)
    return lut[ch]


def rgb_to_lab_u8(img):
    """cv2.COLOR_RGB2Lab on uint8 as its documentation defines it (sRGB transfer function, D65, L * 255 / 100, a + 128, b + 128), in float32
    (OpenCV's 8-bit path is a fixed-point approximation of these formulas: unpinned, a level of difference here and there)."""
    f = np.float32
    v = np.asarray(img, np.uint8).astype(f) * f(1.0 / 255.0)
    lin = np.where(v > f(0.04045), np.power((v + f(0.055)) * f(1.0 / 1.055), f(2.4)).astype(f), v * f(1.0 / 12.92)).astype(f)
    r, g, b = lin[..., 0], lin[..., 1], lin[..., 2]
    X = (f(0.412453) * r + f(0.357580) * g + f(0.180423) * b) * f(1.0 / 0.950456)
    Y = f(0.212671) * r + f(0.715160) * g + f(0.072169) * b
    Z = (f(0.019334) * r + f(0.119193) * g + f(0.950227) * b) * f(1.0 / 1.088754)
    fn = lambda t: np.where(t > f(0.008856), np.cbrt(t).astype(f), f(7.787) * t + f(16.0 / 116.0)).astype(f)
    fx, fy, fz = fn(X), fn(Y), fn(Z)
    L = np.where(Y > f(0.008856), f(116.0) * fy - f(16.0), f(903.3) * Y).astype(f)
    out = np.stack([np.rint(L * f(2.55)), np.rint(f(500.0) * (fx - fy) + f(128.0)), np.rint(f(200.0) * (fy - fz) + f(128.0))], -1)
    return np.clip(out, 0, 255).astype(np.uint8)


def lab_to_rgb_u8(lab):
    f = np.float32
    lab = np.asarray(lab, np.uint8).astype(f)
    l, a, b = lab[..., 0] * f(100.0 / 255.0), lab[..., 1] - f(128), lab[..., 2] - f(128)
    low = l <= f(8.0)
    Y_low = l * f(1.0 / 903.3)
    fy = np.where(low, f(7.787) * Y_low + f(16.0 / 116.0), (l + f(16.0)) * f(1.0 / 116.0)).astype(f)
    Y = np.where(low, Y_low, fy * fy * fy).astype(f)
    fx, fz = a * f(1.0 / 500.0) + fy, fy - b * f(1.0 / 200.0)
    inv = lambda t: np.where(t <= f(0.2068966), (t - f(16.0 / 116.0)) * f(1.0 / 7.787), t * t * t).astype(f)
    X, Z = inv(fx) * f(0.950456), inv(fz) * f(1.088754)
    r = f(3.240479) * X - f(1.53715) * Y - f(0.498535) * Z
    g = f(-0.969256) * X + f(1.875991) * Y + f(0.041556) * Z
    bl = f(0.055648) * X - f(0.204043) * Y + f(1.057311) * Z

    def gam(v):
        v = np.clip(v, f(0), f(1)).astype(f)
        return np.where(v > f(0.0031308), f(1.055) * np.power(v, f(1.0 / 2.4)).astype(f) - f(0.055), f(12.92) * v).astype(f)

    out = np.stack([np.rint(gam(r) * f(255)), np.rint(gam(g) * f(255)), np.rint(gam(bl) * f(255))], -1)
    return np.clip(out, 0, 255).astype(np.uint8)


def clahe_cv(channel, clip, tn):
    """cv2.createCLAHE(clipLimit=clip, tileGridSize=(tn, tn)).apply(channel) on one uint8 channel (modules/imgproc/src/clahe.cpp): pad to
    whole tiles (BORDER_REFLECT_101; BOTH sides get a remainder's worth when either does not divide), per tile clip the histogram at
    max(1, int(clip * area / 256)), spread the excess (whole batches over all bins, the residual one by one at a stride), cumulative
    table * 255 / area rounded; every pixel interpolates the tables of its four nearest tiles."""
    f = np.float32
    ch = np.asarray(channel, np.uint8)
    H, W = ch.shape
    whole = W % tn == 0 and H % tn == 0
    We, He = (W, H) if whole else (W + (tn - W % tn), H + (tn - H % tn))
    tw, th = We // tn, He // tn
    area = tw * th
    ys = np.array([reflect101(np.array([y]), H)[0] for y in range(He)])
    xs = np.array([reflect101(np.array([x]), W)[0] for x in range(We)])
    ext = ch[ys][:, xs]
    climit = max(1, int(f(clip) * f(area) / f(256.0))) if clip > 0 else 0
    scale = f(255.0) / f(area)
    luts = np.zeros((tn, tn, 256), np.uint8)
    for ty in range(tn):
        for tx in range(tn):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].reshape(-1), minlength=256).astype(np.int64)
            if climit > 0:
                clipped = int(np.maximum(hist - climit, 0).sum())
                hist = np.minimum(hist, climit)
                batch = clipped // 256
                residual = clipped - batch * 256
                hist = hist + batch
                if residual:
                    step = max(256 // residual, 1)
                    idx = np.arange(0, 256, step)[:residual]
                    hist[idx] += 1
            luts[ty, tx] = np.clip(np.rint(np.cumsum(hist).astype(f) * scale), 0, 255).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    tyf = yy.astype(f) * (f(1.0) / f(th)) - f(0.5)
    txf = xx.astype(f) * (f(1.0) / f(tw)) - f(0.5)
    ty1, tx1 = np.floor(tyf).astype(np.int64), np.floor(txf).astype(np.int64)
    ya, xa = (tyf - ty1.astype(f)).astype(f), (txf - tx1.astype(f)).astype(f)
    ty2, tx2 = np.minimum(ty1 + 1, tn - 1), np.minimum(tx1 + 1, tn - 1)
    ty1, tx1 = np.maximum(ty1, 0), np.maximum(tx1, 0)
    v = ch.astype(np.int64)
    r0 = luts[ty1, tx1, v].astype(f) * (f(1) - xa) + luts[ty1, tx2, v].astype(f) * xa
    r1 = luts[ty2, tx1, v].astype(f) * (f(1) - xa) + luts[ty2, tx2, v].astype(f) * xa
    return np.clip(np.rint(r0 * (f(1) - ya) + r1 * ya), 0, 255).astype(np.uint8)


def contrast_member(p, img):
    """One member of the `contrast` group (:177-186): imgaug builds a 256-entry table in float32 and casts it to uint8 (truncation)."""
    f = np.float32
    op, d0, d1 = int(p[P_D]), f(p[P_D + 1]), f(p[P_D + 2])
    img = np.asarray(img, np.uint8)
    v = img.astype(f)
    u = v * f(1.0 / 255.0)
    if op == 1:
        return _trunc_u8(f(255) * np.power(u, d0).astype(f))
    if op == 2:
        return _trunc_u8(f(127) + d0 * (v - f(127)))
    if op == 3:
        return _trunc_u8(f(255) / (f(1) + np.exp(d0 * (d1 - u)).astype(f)))
    if op == 4:
        return _trunc_u8(f(255) * d0 * np.log2(f(1) + u).astype(f))
    if op == 6:
        return np.stack([equalize_hist_cv(img[..., k]) for k in range(3)], -1)
    if op == 8:
        return np.stack([clahe_cv(img[..., k], float(d0), int(d1)) for k in range(3)], -1)
    if op in (5, 7):               # the L channel of 8-bit Lab (HistogramEqualization / CLAHE)
        lab = rgb_to_lab_u8(img)
        lab[..., 0] = equalize_hist_cv(lab[..., 0]) if op == 5 else clahe_cv(lab[..., 0], float(d0), int(d1))
        return lab_to_rgb_u8(lab)
    return img


def cloud_blend(cur_u8, alpha, intensity):
    """imgaug's CloudLayer.draw_on_image (augmenters/weather.py; augmentation_pipelines.py:192-193 Fog / Clouds) on a uint8 image:
    clip((1 - alpha) * image + alpha * intensity, 0, 255).astype(uint8) in float32, alpha / intensity [H,W] broadcast over channels."""
    f = np.float32
    a = np.asarray(alpha, f)[..., None]
    it = np.asarray(intensity, f)[..., None]
    v = (f(1) - a) * np.asarray(cur_u8, np.uint8).astype(f) + a * it
    return _trunc_u8(v)


def snow_blend(cur_u8, add, floor_):
    """imgaug's SnowflakesLayer blend on a uint8 image: by sum with `add` (clipped to 0..255), then by maximum with `floor_`; rounded."""
    f = np.float32
    v = np.clip(np.asarray(cur_u8, np.uint8).astype(f) + np.asarray(add, f)[..., None], f(0), f(255))
    return _round_u8(np.maximum(v, np.asarray(floor_, f)[..., None]))


def staged_source(p, src_u8, overlay=None):
    """augment_spatial_kernel: the reference's chain on one (sample, view) - [leading Invert] -> one `arithmetic` member -> one
    `color` member -> one `Blur` member -> one `contrast` member -> the cloud layers of a `weather` member (overlay: fp16
    [layers, 2, H, W], the row names its first layer and their number), uint8 [H,W,3] -> uint8 [H,W,3] (rounded between the groups)."""
    cur = np.asarray(src_u8, dtype=np.uint8)
    if p[P_PREINV] != 0:
        cur = 255 - cur
    op = int(p[P_A])
    if op == 13:
        cur = jpeg_roundtrip(cur, int(p[P_A + 1]))
    elif op == 14:
        cur = filter3(cur, p[P_AK:P_AK + 9])
    elif op == 15:
        cur = filter3(cur, p[P_AK:P_AK + 9], pil=True, offset=float(p[P_A + 1]), scale=float(p[P_A + 2]) or 1.0)
    elif op != 0:
        cur = arith_pointwise(p, cur)
    cur = colour_member(p, cur)
    mode = int(p[P_C])
    if mode == 1:
        cur = filter7(cur, np.asarray(p[P_KERN:P_KERN + 49], dtype=np.float32).reshape(7, 7))
    elif mode == 2:
        cur = median_blur(cur, int(p[P_C + 1]))
    elif mode == 3:
        cur = bilateral_blur(cur, int(p[P_C + 1]), float(p[P_C + 2]), float(p[P_C + 3]))
    cur = contrast_member(p, cur)
    n, first, snow = int(p[P_W]), int(p[P_W + 1]), int(p[P_W + 2])
    if overlay is not None and n > 0:
        for l in range(first, first + n):
            cur = snow_blend(cur, overlay[l, 0], overlay[l, 1]) if snow else cloud_blend(cur, overlay[l, 0], overlay[l, 1])
    return cur


def augment_views(img, params, theta, mean, std, overlay=None, warp_maps=None):
    """img uint8 [B,H,W,3], params [B,2,96], theta [B,3,3] -> fp32 [B,3,3,H,W].  warp_maps fp32 [maps,2,H,W]: a sample whose view-2
    row has params[84] = m > 0 is sampled at map m - 1's (x, y) source positions instead of theta's (imgaug PiecewiseAffine through
    skimage.transform.warp(order=1, mode="constant", cval=0): bilinear weights, zero beyond the image)."""
    f = np.float32
    img = np.asarray(img)
    B, H, W, _ = img.shape
    mean, istd = np.asarray(mean, f), (f(1) / np.asarray(std, f)).astype(f)
    out = np.zeros((B, 3, 3, H, W), f)
    ys, xs = np.mgrid[0:H, 0:W]

    def norm(c):          # [H,W,3] -> [3,H,W]
        return ((c * f(1.0 / 255.0) - mean) * istd).astype(f).transpose(2, 0, 1)

    for b in range(B):
        out[b, 0] = norm(img[b].astype(f))
        out[b, 1] = norm(staged_source(params[b, 0], img[b], overlay).astype(f))
        th = theta[b].astype(f)
        xn = f(2) * xs.astype(f) / f(W - 1) - f(1)
        yn = f(2) * ys.astype(f) / f(H - 1) - f(1)
        sx = ((th[0, 0] * xn + th[0, 1] * yn + th[0, 2]) + f(1)) * f(0.5) * f(W - 1)
        sy = ((th[1, 0] * xn + th[1, 1] * yn + th[1, 2]) + f(1)) * f(0.5) * f(H - 1)
        m = int(params[b, 1, 84]) if warp_maps is not None else 0
        if 0 < m <= len(warp_maps):
            sx, sy = warp_maps[m - 1, 0].astype(f), warp_maps[m - 1, 1].astype(f)
        x0, y0 = np.floor(sx), np.floor(sy)
        ax, ay = (sx - x0).astype(f), (sy - y0).astype(f)
        x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
        col2 = staged_source(params[b, 1], img[b], overlay).astype(f)
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


def directed_edge_kernel(alpha, direction):
    """iaa.DirectedEdgeDetect(alpha, direction): the 8 neighbours weighted by (1 - angle to the direction / 180 deg)^4, normalised,
    negated, centre 1; blended with the identity kernel by alpha."""
    rad = np.deg2rad(float(direction) * 360.0 % 360.0)
    dvec = np.array([np.cos(rad - 0.5 * np.pi), np.sin(rad - 0.5 * np.pi)])
    m = np.zeros((3, 3), np.float64)
    for x in (-1, 0, 1):
        for y in (-1, 0, 1):
            if (x, y) != (0, 0):
                cell = np.array([x, y], np.float64)
                cosang = np.clip(cell @ dvec / (np.linalg.norm(cell) * np.linalg.norm(dvec)), -1.0, 1.0)
                m[y + 1, x + 1] = (1.0 - np.rad2deg(np.arccos(cosang)) / 180.0) ** 4
    m = -m / m.sum()
    m[1, 1] = 1.0
    ident = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], np.float64)
    return ((1.0 - alpha) * ident + alpha * m).astype(np.float32)


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
