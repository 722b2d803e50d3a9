"""Per-sample augmentation parameters for the device view maker (csrc/kernels/datapipe.h::augment_views_kernel).

Geometry = the reference's `iaa.Affine(scale 0.6-1.1, translate +-2 %, rotate +-10 deg, shear x +-45 / y +-10 deg)` applied
to view 2 with probability 0.7 (datasetsupervised_kmeans.py:40-45,60), and `theta` derived from the pixel matrix exactly as
:65-71 does: theta = W_ . M_px^-1 . W_^-1 with W_ = [[2/(w-1),0,-1],[0,2/(h-1),-1],[0,0,1]] (the dataset's W_inv . M . W
factors cancel when the augmentation runs at the network resolution, which it does here).
Colour = the POINTWISE members of the imgaug pipelines the configs select (augmentation_pipelines.py: severity 5 for view
`augment_tfs`, 6 for `augment_color`): invert, grayscale blend, channel shuffle, gamma / linear contrast, brightness and
per-channel gains (MultiplyBrightness, ChangeColorTemperature), solarize, additive gaussian / multiplicative / impulse
noise, with the reference's ranges and its "identity with probability 0.2 / each group with probability 0.7" structure.
The 3x3-support SPATIAL members are reproduced as one 3x3 filter in front of the pointwise chain (GaussianBlur / AverageBlur as
their 3x3 truncation, Sharpen, Emboss, EdgeDetect with imgaug's effect matrices); median / motion / bilateral blur, weather,
JPEG and elastic members are not.
"""
from __future__ import annotations

import math

import numpy as np

AUG_NP = 32
IDENTITY_PARAMS = np.array([0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 256, 0, 0, 0] + [0, 0, 0, 0, 1, 0, 0, 0, 0] + [0] * 7,
                           dtype=np.float32)            # [16:25] = the 3x3 filter (identity), [14] = filter on
_NOCHANGE = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=np.float64)


def _set_filter(p, kernel):
    p[14] = 1.0
    p[16:25] = np.asarray(kernel, dtype=np.float32).reshape(-1)


def _blend(alpha, effect):
    """imgaug's convolutional augmenters: (1 - alpha) * identity + alpha * effect matrix."""
    return (1.0 - alpha) * _NOCHANGE + alpha * np.asarray(effect, dtype=np.float64)


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


def sample_theta(rs: np.random.RandomState, batch: int, h: int, w: int, p_warp: float = 0.7) -> np.ndarray:
    out = np.tile(np.eye(3, dtype=np.float32), (batch, 1, 1))
    for b in range(batch):
        if rs.uniform() > 1.0 - p_warp:           # `random.random() > 0.3` (:60)
            out[b] = theta_from_pixel_matrix(affine_pixel_matrix(rs, h, w), h, w)
    return out


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
        k = rs.randint(0, 8)
        if k == 0:
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
    if severity == 5 and p[14] == 0 and rs.uniform() < 0.7:            # `Blur`: Sharpen | one of the blurs
        if rs.uniform() < 0.5:
            a, light = rs.uniform(0.0, 0.5), rs.uniform(0.0, 0.5)
            _set_filter(p, _blend(a, [[-1, -1, -1], [-1, 8 + light, -1], [-1, -1, -1]]))
        elif rs.uniform() < 0.5:                  # GaussianBlur(sigma 0.5 - 1.5), truncated to 3 x 3
            sg = rs.uniform(0.5, 1.5)
            g1 = np.array([math.exp(-0.5 / sg ** 2), 1.0, math.exp(-0.5 / sg ** 2)])
            g1 /= g1.sum()
            _set_filter(p, np.outer(g1, g1))
        else:                                     # AverageBlur, 3 x 3
            _set_filter(p, np.full((3, 3), 1.0 / 9.0))
    if groups["contrast"] is not None and rs.uniform() < groups["contrast"]:
        k = rs.randint(0, 3)
        if k == 0:
            p[3] *= rs.uniform(0.5, 2.0)
        elif k == 1:
            p[7] = rs.uniform(0.5, 1.0)           # LinearContrast
        # k == 2: histogram members (equalisation / CLAHE): spatial statistics, not reproduced
    return p


def sample_colour_params(rs: np.random.RandomState, batch: int, severity: int = 5) -> np.ndarray:
    """fp32 [batch, 2, 32]: view 1 from pipeline `severity`, view 2 from the same pipeline (both come from `augment_tfs`,
    datasetsupervised_kmeans.py:57)."""
    out = np.empty((batch, 2, AUG_NP), dtype=np.float32)
    for b in range(batch):
        out[b, 0] = _colour_params(rs, severity)
        out[b, 1] = _colour_params(rs, severity)
    return out
