"""Per-sample augmentation parameters for the device view maker (csrc/kernels/datapipe.h: augment_spatial_kernel +
augment_views_kernel).

Geometry = the reference's `iaa.Affine(scale 0.6-1.1, translate +-2 %, rotate +-10 deg, shear x +-45 / y +-10 deg)` applied
to view 2 with probability 0.7 (datasetsupervised_kmeans.py:40-45,60), and `theta` derived from the pixel matrix exactly as
:65-71 does: theta = W_ . M_px^-1 . W_^-1 with W_ = [[2/(w-1),0,-1],[0,2/(h-1),-1],[0,0,1]] (the dataset's W_inv . M . W
factors cancel when the augmentation runs at the network resolution, which it does here).  A sample whose 0.7 draw fails
gets the PLAIN image as view 2 (:72-74): its colour parameters are the identity.
Colour = the members of the imgaug pipelines the configs select (augmentation_pipelines.py: severity 5 for `augment_tfs`):
* pointwise (augment_views_kernel): invert, grayscale blend, channel shuffle, gamma / linear contrast, brightness and
  per-channel gains (MultiplyBrightness, ChangeColorTemperature), solarize, additive gaussian / multiplicative / impulse
  noise, with the reference's ranges and its "identity with probability 0.2 / each group with probability 0.7" structure;
* neighbourhood (augment_spatial_kernel, a pre-pass): JpegCompression(70-99) as PIL's baseline JPEG round trip; the `Blur`
  group as imgaug builds it - Sharpen | OneOf[GaussianBlur(0.5-1.5) with its 5 x 5 kernel, AverageBlur(k 2-6) with cv2.blur's
  anchor, MedianBlur(k 3-7), MotionBlur(k = 5, any angle / direction), BilateralBlur(d 3-10, sigmas 10-250)]; Emboss and
  EdgeDetect with imgaug's effect matrices.
NOT reproduced (the draw that would select them leaves the image unchanged): the weather members (Fog, Clouds, Snowflakes,
Rain :191-196), histogram equalisation / CLAHE, k-means / uniform colour quantisation, hue / saturation arithmetic in HSV,
CoarseDropout, the Laplace / Poisson noise variants, DirectedEdgeDetect and the PIL filter presets; severity 2's
ElasticTransformation / PerspectiveTransform (no shipped config selects severity 2).  INTEGRATION.md lists them.
"""
from __future__ import annotations

import math

import numpy as np

AUG_NP = 96
P_MODE, P_K, P_JPEG, P_SIGC, P_SIGS, P_KERN = 14, 15, 25, 26, 27, 32        # kernels/datapipe.h
IDENTITY_PARAMS = np.zeros(AUG_NP, dtype=np.float32)
IDENTITY_PARAMS[3:8] = 1.0                                                    # gamma, three gains, contrast alpha
IDENTITY_PARAMS[12] = 256.0                                                   # solarize threshold: off
_NOCHANGE = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=np.float64)


def _set_filter(p, kernel, anchor=None):
    """A correlation kernel (odd or even side <= 7) into the 7 x 7 grid; `anchor` = (row, col) of the kernel cell that sits on
    the pixel (cv2's default: side // 2)."""
    k = np.asarray(kernel, dtype=np.float64)
    ay, ax = anchor if anchor is not None else (k.shape[0] // 2, k.shape[1] // 2)
    grid = np.zeros((7, 7), dtype=np.float64)
    grid[3 - ay:3 - ay + k.shape[0], 3 - ax:3 - ax + k.shape[1]] = k
    p[P_MODE] = 1.0
    p[P_KERN:P_KERN + 49] = grid.reshape(-1).astype(np.float32)


def _blend(alpha, effect):
    """imgaug's convolutional augmenters: (1 - alpha) * identity + alpha * effect matrix."""
    return (1.0 - alpha) * _NOCHANGE + alpha * np.asarray(effect, dtype=np.float64)


def gaussian_kernel5(sigma):
    """cv2.getGaussianKernel(5, sigma) x its transpose: what iaa.GaussianBlur(sigma <= 1.5) convolves with (imgaug picks
    ksize = max(3.3 sigma, 5) for sigma < 3)."""
    g = np.exp(-0.5 * (np.arange(5) - 2.0) ** 2 / sigma ** 2)
    g /= g.sum()
    return np.outer(g, g)


def motion_kernel(k, angle_deg, direction):
    """iaa.MotionBlur(k, angle, direction, order=1): the middle column of a k x k matrix holds linspace(d, 1 - d),
    d = (direction + 1) / 2; the uint8 matrix is rotated by `angle` about its centre (bilinear, zero padding) and divided by
    its sum."""
    k = k if k % 2 else k + 1
    d = (min(max(float(direction), -1.0), 1.0) + 1.0) / 2.0
    line = np.floor(np.linspace(d, 1.0 - d, num=k) * 255.0)
    c, ang = (k - 1) / 2.0, math.radians(angle_deg)
    cs, sn = math.cos(ang), math.sin(ang)
    out = np.zeros((k, k), dtype=np.float64)
    for y in range(k):
        for x in range(k):
            xs, ys = cs * (x - c) + sn * (y - c) + c, -sn * (x - c) + cs * (y - c) + c
            x0, y0 = math.floor(xs), math.floor(ys)
            v = 0.0
            for yy, wy in ((y0, 1.0 - (ys - y0)), (y0 + 1, ys - y0)):
                for xx, wx in ((x0, 1.0 - (xs - x0)), (x0 + 1, xs - x0)):
                    if 0 <= yy < k and xx == k // 2:
                        v += wy * wx * line[yy]
            out[y, x] = math.floor(v + 0.5)
    total = out.sum()
    if total <= 0:
        out[:, k // 2] = 1.0
        total = float(k)
    return out / total


def affine_pixel_matrix(rs: np.random.RandomState, h: int, w: int) -> np.ndarray:
    """Forward pixel-space matrix (source px -> warped px) of one iaa.Affine draw, centred like imgaug does."""
    sx, sy = rs.uniform(0.6, 1.1, size=2)
    tx, ty = rs.uniform(-0.02, 0.02, size=2) * (w, h)
    rot = math.radians(rs.uniform(-10.0, 10.0))
    shx = math.radians(rs.uniform(-45.0, 45.0))
    shy = math.radians(rs.uniform(-10.0, 10.0))
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    to_origin = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]], dtype=np.float64)
    from_origin = np.array([[1, 0, cx + tx], [0, 1, cy + ty], [0, 0, 1]], dtype=np.float64)
    scale = np.diag([sx, sy, 1.0])
    shear = np.array([[1, math.tan(shx), 0], [math.tan(shy), 1, 0], [0, 0, 1]], dtype=np.float64)
    rotm = np.array([[math.cos(rot), -math.sin(rot), 0], [math.sin(rot), math.cos(rot), 0], [0, 0, 1]])
    return from_origin @ rotm @ shear @ scale @ to_origin


def theta_from_pixel_matrix(forward_px: np.ndarray, h: int, w: int) -> np.ndarray:
    """datasetsupervised_kmeans.py:69-70 at network resolution: theta = W_ . M^-1 . W_^-1 (float32 like the dataset)."""
    norm = np.array([[2.0 / (w - 1), 0, -1], [0, 2.0 / (h - 1), -1], [0, 0, 1]], dtype=np.float64)
    return (norm @ np.linalg.inv(forward_px) @ np.linalg.inv(norm)).astype(np.float32)


def sample_theta(rs: np.random.RandomState, batch: int, h: int, w: int, p_warp: float = 0.7, return_warped: bool = False):
    """theta fp32 [batch, 3, 3]; with return_warped also the bool [batch] of samples whose view 2 really is warped."""
    out = np.tile(np.eye(3, dtype=np.float32), (batch, 1, 1))
    warped = np.zeros(batch, dtype=bool)
    for b in range(batch):
        if rs.uniform() > 1.0 - p_warp:           # `random.random() > 0.3` (:60)
            out[b] = theta_from_pixel_matrix(affine_pixel_matrix(rs, h, w), h, w)
            warped[b] = True
    return (out, warped) if return_warped else out


def _colour_params(rs: np.random.RandomState, severity: int) -> np.ndarray:
    p = IDENTITY_PARAMS.copy()
    p[13] = rs.randint(0, 1 << 24)
    if severity <= 0:
        return p
    if severity == 5:
        if rs.uniform() < 0.2:                    # iaa.Sometimes(0.2, Identity, Sequential[...])
            return p
        groups = {"arith": 1.0, "color": 0.7, "contrast": 0.7}
    else:
        groups = {"arith": None, "color": 1.0, "contrast": None}      # severity 6 / others: OneOf over colour members
    if groups["arith"] is not None:               # `arithmetic`: always one member
        k = rs.randint(0, 9)
        if k == 8:                                # JpegCompression(compression 70-99) -> PIL quality 31 .. 2
            comp = rs.uniform(70, 99)
            p[P_JPEG] = float(np.clip(np.round(1 + 99 * (1.0 - comp / 100.0)), 1, 100))
        elif k == 0:
            p[8] = rs.uniform(-40, 40)
        elif k == 1:
            p[9] = rs.uniform(0, 0.2 * 255)
        elif k == 2:
            p[4:7] = rs.uniform(0.5, 1.5) if rs.uniform() < 0.5 else rs.uniform(0.5, 1.5, size=3)
        elif k == 3:
            p[10] = 0.5
        elif k == 4:
            p[11] = 0.1
        elif k == 5:
            p[0] = float(rs.uniform() < 0.15)
        elif k == 6:
            if rs.uniform() < 0.5:
                p[12] = rs.uniform(32, 128)
        elif k == 7:                              # Emboss(alpha 0-1, strength 0.5-1.5) / EdgeDetect(alpha 0-1)
            a = rs.uniform(0.0, 1.0)
            if rs.uniform() < 0.5:
                st = rs.uniform(0.5, 1.5)
                _set_filter(p, _blend(a, [[-1 - st, 0 - st, 0], [0 - st, 1, 0 + st], [0, 0 + st, 1 + st]]))
            else:
                _set_filter(p, _blend(a, [[0, 1, 0], [1, -4, 1], [0, 1, 0]]))
    if groups["color"] is not None and rs.uniform() < groups["color"]:
        k = rs.randint(0, 5)
        if k == 0:
            p[4:7] *= rs.uniform(0.5, 1.5)        # MultiplyAndAddToBrightness / MultiplyBrightness
            p[8] += rs.uniform(-30, 30) if severity == 5 else 0.0
        elif k == 1:
            p[1] = rs.uniform(0.0, 1.0)           # Grayscale(alpha)
        elif k == 2:
            if rs.uniform() < 0.35:
                p[2] = rs.randint(0, 6)           # ChannelShuffle(0.35)
        elif k == 3:                              # ChangeColorTemperature(1100 .. 10000 K): warm <-> cool channel gains
            t = rs.uniform(-1.0, 1.0)
            p[4] *= 1.0 + 0.25 * t
            p[6] *= 1.0 - 0.25 * t
        elif k == 4:
            p[3] = rs.uniform(0.5, 2.0)           # GammaContrast
    if severity == 5 and p[P_MODE] == 0 and rs.uniform() < 0.7:        # `Blur`: OneOf[Sharpen, OneOf[five blurs]] (:165-176)
        if rs.uniform() < 0.5:
            a, light = rs.uniform(0.0, 0.5), rs.uniform(0.0, 0.5)
            _set_filter(p, _blend(a, [[-1, -1, -1], [-1, 8 + light, -1], [-1, -1, -1]]))
        else:
            m = rs.randint(0, 5)
            if m == 0:                            # GaussianBlur(sigma 0.5 - 1.5): 5 x 5
                _set_filter(p, gaussian_kernel5(rs.uniform(0.5, 1.5)))
            elif m == 1:                          # AverageBlur(k 2 - 6): cv2.blur, anchor k // 2
                k = rs.randint(2, 7)
                _set_filter(p, np.full((k, k), 1.0 / (k * k)))
            elif m == 2:                          # MedianBlur(k 3 - 7): even draws go to the next odd size
                k = rs.randint(3, 8)
                p[P_MODE], p[P_K] = 2.0, float(k if k % 2 else k + 1)
            elif m == 3:                          # MotionBlur(k = 5, angle 0 - 360, direction -1 .. 1)
                _set_filter(p, motion_kernel(5, rs.uniform(0.0, 360.0), rs.uniform(-1.0, 1.0)))
            else:                                 # BilateralBlur(d 3 - 10, sigma_color / sigma_space 10 - 250)
                p[P_MODE], p[P_K] = 3.0, float(rs.randint(3, 11))
                p[P_SIGC], p[P_SIGS] = rs.uniform(10, 250), rs.uniform(10, 250)
    if groups["contrast"] is not None and rs.uniform() < groups["contrast"]:
        k = rs.randint(0, 3)
        if k == 0:
            p[3] *= rs.uniform(0.5, 2.0)
        elif k == 1:
            p[7] = rs.uniform(0.5, 1.0)           # LinearContrast
        # k == 2: histogram members (equalisation / CLAHE): spatial statistics, not reproduced
    return p


def sample_colour_params(rs: np.random.RandomState, batch: int, severity: int = 5, warped=None) -> np.ndarray:
    """fp32 [batch, 2, 96]: view 1 from pipeline `severity`, view 2 from the same pipeline (both come from `augment_tfs`,
    datasetsupervised_kmeans.py:57).  `warped` (bool [batch], from sample_theta): a sample whose warp draw failed gets the
    plain image as view 2 (:72-74 `image_view = image`), i.e. identity parameters."""
    out = np.empty((batch, 2, AUG_NP), dtype=np.float32)
    for b in range(batch):
        out[b, 0] = _colour_params(rs, severity)
        out[b, 1] = _colour_params(rs, severity)
        if warped is not None and not warped[b]:
            out[b, 1] = IDENTITY_PARAMS
            out[b, 1, 13] = rs.randint(0, 1 << 24)
    return out


# ------------------------------------------------------------------------------------------------ finetuning pipeline
def _finetune_colour_params(rs: np.random.RandomState) -> np.ndarray:
    """One draw of the finetuning pipeline's colour part (Dino/dataset/dataset_pretrain.py:79-146): Sometimes(0.6, Invert(0.1)),
    Sometimes(0.8, OneOf[41 members]), Sometimes(0.6, Blur group without the bilateral member), Sometimes(0.6, contrast group).
    Members this implementation does not reproduce keep their share of the draw and leave the image unchanged."""
    p = IDENTITY_PARAMS.copy()
    p[13] = rs.randint(0, 1 << 24)
    if rs.uniform() < 0.6:
        p[0] = float(rs.uniform() < 0.1)
    if rs.uniform() < 0.8:
        k = rs.randint(0, 41)                     # position in the reference's list (:85-127)
        if k == 0:
            if rs.uniform() < 0.35:
                p[2] = rs.randint(0, 6)           # ChannelShuffle(0.35)
        elif k == 1:
            p[8] = rs.uniform(-40, 40)            # AddElementwise, drawn per image here
        elif k == 2:
            p[9] = rs.uniform(0, 0.2 * 255)       # AdditiveGaussianNoise
        elif k == 5 or 11 <= k <= 13:
            p[11] = 0.1                           # ImpulseNoise / SaltAndPepper / Salt / Pepper
        elif k == 6:
            p[4:7] = rs.uniform(0.5, 1.5) if rs.uniform() < 0.5 else rs.uniform(0.5, 1.5, size=3)      # Multiply(per_channel 0.5)
        elif k == 7:
            p[10] = 0.5                           # MultiplyElementwise
        elif k == 14:
            if rs.uniform() < 0.5:
                p[12] = rs.uniform(32, 128)       # Solarize
        elif k == 15:
            p[P_JPEG] = float(np.clip(np.round(1 + 99 * (1.0 - rs.uniform(70, 99) / 100.0)), 1, 100))
        elif k == 16:
            a, st = rs.uniform(0.0, 1.0), rs.uniform(0.5, 1.5)
            _set_filter(p, _blend(a, [[-1 - st, 0 - st, 0], [0 - st, 1, 0 + st], [0, 0 + st, 1 + st]]))
        elif k == 17:
            _set_filter(p, _blend(rs.uniform(0.0, 1.0), [[0, 1, 0], [1, -4, 1], [0, 1, 0]]))
        elif k == 22:
            p[4:7] *= rs.uniform(0.5, 1.5)        # MultiplyBrightness
        elif k == 23:
            p[4:7] *= rs.uniform(0.5, 1.5)        # MultiplyAndAddToBrightness
            p[8] += rs.uniform(-30, 30)
        elif k == 27:
            p[1] = rs.uniform(0.0, 1.0)           # Grayscale
        elif k == 30:
            t = rs.uniform(-1.0, 1.0)             # ChangeColorTemperature
            p[4] *= 1.0 + 0.25 * t
            p[6] *= 1.0 - 0.25 * t
        # others (Laplace / Poisson noise, dropouts, HSV arithmetic, quantisation, edge presets, weather): not reproduced
    if p[P_MODE] == 0 and rs.uniform() < 0.6:
        if rs.uniform() < 0.5:
            a, light = rs.uniform(0.0, 0.5), rs.uniform(0.0, 0.5)
            _set_filter(p, _blend(a, [[-1, -1, -1], [-1, 8 + light, -1], [-1, -1, -1]]))
        else:
            m = rs.randint(0, 4)
            if m == 0:
                _set_filter(p, gaussian_kernel5(rs.uniform(0.5, 1.5)))
            elif m == 1:
                k = rs.randint(2, 7)
                _set_filter(p, np.full((k, k), 1.0 / (k * k)))
            elif m == 2:
                k = rs.randint(3, 8)
                p[P_MODE], p[P_K] = 2.0, float(k if k % 2 else k + 1)
            else:
                _set_filter(p, motion_kernel(5, rs.uniform(0.0, 360.0), rs.uniform(-1.0, 1.0)))
    if rs.uniform() < 0.6:
        k = rs.randint(0, 8)
        if k == 0:
            p[3] *= rs.uniform(0.5, 2.0)          # GammaContrast
        elif k == 1:
            p[7] = rs.uniform(0.5, 1.0)           # LinearContrast
    return p


def sample_finetune_params(rs: np.random.RandomState, batch: int, h: int, w: int):
    """(params fp32 [batch, 2, 96] - only row 1 is used, theta fp32 [batch, 3, 3]) for the finetuning augmentation
    (dataset_pretrain.py:79-158): colour as above, geometry = Sometimes(0.6, OneOf[Affine (the pretraining ranges), PiecewiseAffine
    (not reproduced), Rotate(-45, 45)])."""
    params = np.tile(IDENTITY_PARAMS, (batch, 2, 1)).astype(np.float32)
    theta = np.tile(np.eye(3, dtype=np.float32), (batch, 1, 1))
    for b in range(batch):
        params[b, 1] = _finetune_colour_params(rs)
        if rs.uniform() < 0.6:
            g = rs.randint(0, 3)
            if g == 0:
                theta[b] = theta_from_pixel_matrix(affine_pixel_matrix(rs, h, w), h, w)
            elif g == 2:
                rot = math.radians(rs.uniform(-45.0, 45.0))
                cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
                to_o = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]], dtype=np.float64)
                back = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1]], dtype=np.float64)
                rotm = np.array([[math.cos(rot), -math.sin(rot), 0], [math.sin(rot), math.cos(rot), 0], [0, 0, 1]])
                theta[b] = theta_from_pixel_matrix(back @ rotm @ to_o, h, w)
    return params, theta
