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
AUG_NP = 32


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


def source(p, src):
    """the optionally 3x3-filtered image (borders replicate): src [H,W,3] fp32 -> [H,W,3] fp32."""
    f = np.float32
    if p[14] == 0:
        return src
    H, W, _ = src.shape
    out = np.zeros_like(src)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            yy = np.clip(np.arange(H) + dy, 0, H - 1)
            xx = np.clip(np.arange(W) + dx, 0, W - 1)
            out = out + f(p[16 + 3 * (dy + 1) + (dx + 1)]) * src[yy][:, xx]
    return out.astype(f)


def augment_views(img, params, theta, mean, std):
    """img uint8 [B,H,W,3], params [B,2,32], theta [B,3,3] -> fp32 [B,3,3,H,W]."""
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
        out[b, 1] = norm(colour(params[b, 0], source(params[b, 0], src), pid))
        th = theta[b].astype(f)
        xn = f(2) * xs.astype(f) / f(W - 1) - f(1)
        yn = f(2) * ys.astype(f) / f(H - 1) - f(1)
        sx = ((th[0, 0] * xn + th[0, 1] * yn + th[0, 2]) + f(1)) * f(0.5) * f(W - 1)
        sy = ((th[1, 0] * xn + th[1, 1] * yn + th[1, 2]) + f(1)) * f(0.5) * f(H - 1)
        x0, y0 = np.floor(sx), np.floor(sy)
        ax, ay = (sx - x0).astype(f), (sy - y0).astype(f)
        x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
        col2 = colour(params[b, 1], source(params[b, 1], src), pid)   # colour of every source pixel once
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
