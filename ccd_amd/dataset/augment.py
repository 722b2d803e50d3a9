"""Per-sample augmentation parameters for the device view maker (csrc/kernels/datapipe.h: augment_spatial_kernel +
augment_views_kernel).

Geometry = the reference's `iaa.Affine(scale 0.6-1.1, translate +-2 %, rotate +-10 deg, shear x +-45 / y +-10 deg)` applied
to view 2 with probability 0.7 (datasetsupervised_kmeans.py:40-45,60), and `theta` derived from the pixel matrix exactly as
:65-71 does: theta = W_ . M_px^-1 . W_^-1 with W_ = [[2/(w-1),0,-1],[0,2/(h-1),-1],[0,0,1]] (the dataset's W_inv . M . W
factors cancel when the augmentation runs at the network resolution, which it does here).  A sample whose 0.7 draw fails
gets the PLAIN image as view 2 (:72-74): its colour parameters are the identity.

Colour = the reference's imgaug chain: its member LISTS and probabilities, every member of the pretraining chain (50) and of the
finetuning list reproduced (the members the image lacks the library for are restated from the published algorithms and say so:
INTEGRATION.md) (augmentation_pipelines.py:120-205,
severity 5 - what the shipped pretraining configs select; dataset_pretrain.py:79-158 for finetuning): `Sometimes(0.2, Identity,
Sequential[arithmetic: OneOf 21, color: Sometimes(0.7, OneOf 9), Blur: Sometimes(0.7, ...), contrast: Sometimes(0.7, OneOf 8),
weather: Sometimes(0.7, OneOf 4)])`.  A draw picks the member by its POSITION in the reference's list and writes what the kernel
needs for it (one member per group, applied in the reference's order on a uint8 image).  Reproduced: AddElementwise, the
Gaussian / Laplace / Poisson noises, Multiply(Elementwise), Dropout, CoarseDropout, Dropout2d, ImpulseNoise / SaltAndPepper /
Salt / Pepper, Invert, Solarize, JpegCompression, Emboss, EdgeDetect, DirectedEdgeDetect, pillike.FilterEdgeEnhanceMore /
FilterContour; the HSV hue shifts, MultiplyAndAddToBrightness (in RGB), MultiplyHueAndSaturation, AddToHueAndSaturation,
Grayscale, UniformColorQuantization, ChangeColorTemperature (as channel gains); Sharpen and the five blurs; Gamma / Linear /
Sigmoid / Log contrast, AllChannelsHistogramEqualization; Fog and Clouds (round 5: cloud layers drawn on the host - weather.py - and
blended on the device).
HistogramEqualization / CLAHE (the L channel of 8-bit Lab, float formulas) and AllChannelsCLAHE (round 5: OpenCV's tile algorithm).
KMeansColorQuantization (round 5: Lloyd's iteration on the Lab triples, cv2.kmeans' rules).
Snowflakes and Rain (round 5: salt noise on a shrunk canvas, gated, up-sampled, blurred, motion-smeared - on the host like the clouds).
PiecewiseAffine (finetuning geometry, round 5): a dense source-position map per sample drawn on the host (weather.piecewise_affine_map).
NOT reproduced - the draw that selects it leaves the image unchanged: severity 2's ElasticTransformation / PerspectiveTransform (no shipped config).
"""
from __future__ import annotations

import math

import numpy as np

AUG_NP = 96
P_SEED, P_PREINV, P_A, P_AK, P_B, P_C, P_D, P_KERN, P_W = 0, 1, 2, 9, 18, 24, 28, 32, 81    # kernels/datapipe.h
A_ADD_ELEM, A_GAUSS, A_LAPLACE, A_POISSON, A_MUL, A_MUL_ELEM, A_DROPOUT, A_COARSE, A_DROP2D, A_REPLACE, A_INVERT, A_SOLARIZE, \
    A_JPEG, A_FILTER, A_PILFILTER = range(1, 16)
B_HUE_ADD, B_BRIGHT, B_MUL_HS, B_ADD_HS, B_GRAY, B_KMEANS, B_UNIFORM_Q, B_GAINS, B_SHUFFLE = range(1, 10)
C_FILTER, C_MEDIAN, C_BILATERAL = 1, 2, 3
D_GAMMA, D_LINEAR, D_SIGMOID, D_LOG, D_HISTEQ_LAB, D_HISTEQ_ALL, D_CLAHE_LAB, D_CLAHE_ALL = 1, 2, 3, 4, 5, 6, 7, 8
IDENTITY_PARAMS = np.zeros(AUG_NP, dtype=np.float32)
_NOCHANGE = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=np.float64)


def _set_filter(p, kernel, anchor=None):
    """`Blur` group: a correlation kernel (odd or even side <= 7) into the 7 x 7 grid; `anchor` = (row, col) of the kernel cell
    that sits on the pixel (cv2's default: side // 2)."""
    k = np.asarray(kernel, dtype=np.float64)
    ay, ax = anchor if anchor is not None else (k.shape[0] // 2, k.shape[1] // 2)
    grid = np.zeros((7, 7), dtype=np.float64)
    grid[3 - ay:3 - ay + k.shape[0], 3 - ax:3 - ax + k.shape[1]] = k
    p[P_C] = C_FILTER
    p[P_KERN:P_KERN + 49] = grid.reshape(-1).astype(np.float32)


def _set_arith_filter(p, kernel3, pil=False, offset=0.0, scale=1.0):
    """`arithmetic` group: a 3 x 3 correlation - imgaug's Convolve members (cv2.filter2D, reflect-101 border) or a PIL
    ImageFilter.Kernel (sum / scale + offset, border pixels copied)."""
    p[P_A] = A_PILFILTER if pil else A_FILTER
    p[P_A + 1], p[P_A + 2] = offset, scale
    p[P_AK:P_AK + 9] = np.asarray(kernel3, dtype=np.float64).reshape(-1).astype(np.float32)


def _blend(alpha, effect):
    """imgaug's convolutional augmenters: (1 - alpha) * identity + alpha * effect matrix."""
    return (1.0 - alpha) * _NOCHANGE + alpha * np.asarray(effect, dtype=np.float64)


def directed_edge_kernel(alpha, direction):
    """iaa.DirectedEdgeDetect: the 8 neighbours weighted by (1 - angle to the direction / 180 deg)^4 (direction 0 = up, clockwise
    in image coordinates), normalised, negated, centre 1; blended with the identity kernel by alpha."""
    rad = math.radians((direction * 360.0) % 360.0)
    dx, dy = math.cos(rad - 0.5 * math.pi), math.sin(rad - 0.5 * math.pi)
    m = np.zeros((3, 3), dtype=np.float64)
    for x in (-1, 0, 1):
        for y in (-1, 0, 1):
            if (x, y) != (0, 0):
                cosang = (x * dx + y * dy) / (math.hypot(x, y) * math.hypot(dx, dy))
                m[y + 1, x + 1] = (1.0 - math.degrees(math.acos(min(1.0, max(-1.0, cosang)))) / 180.0) ** 4
    m = -m / m.sum()
    m[1, 1] = 1.0
    return _blend(alpha, m)


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


# ------------------------------------------------------------------------------------------------ the member lists
def _per_channel(rs, prob):
    return float(rs.uniform() < prob)


def _jpeg_quality(rs):
    """JpegCompression(compression 70-99) -> PIL quality 31 .. 2 (imgaug: 1 + 99 * (1 - compression / 100), rounded, clipped)."""
    return float(np.clip(np.round(1 + 99 * (1.0 - rs.uniform(70, 99) / 100.0)), 1, 100))


def _arith_member(p, rs, name, h, w):
    """One member of the `arithmetic` list (augmentation_pipelines.py:122-144 = dataset_pretrain.py:86-106)."""
    a = P_A
    if name == "AddElementwise":                  # integers -40 .. 40, one per pixel
        p[a], p[a + 1] = A_ADD_ELEM, 40
    elif name == "AdditiveGaussianNoise":
        p[a], p[a + 1] = A_GAUSS, rs.uniform(0, 0.2 * 255)
    elif name == "AdditiveLaplaceNoise":
        p[a], p[a + 1] = A_LAPLACE, rs.uniform(0, 0.2 * 255)
    elif name == "AdditivePoissonNoise":
        p[a], p[a + 1] = A_POISSON, rs.uniform(0, 40)
    elif name == "Multiply":                      # per_channel = 0.5
        p[a] = A_MUL
        p[a + 1:a + 4] = rs.uniform(0.5, 1.5, size=3) if rs.uniform() < 0.5 else rs.uniform(0.5, 1.5)
    elif name == "MultiplyElementwise":
        p[a], p[a + 1], p[a + 3], p[a + 2] = A_MUL_ELEM, 0.5, 1.5, _per_channel(rs, 0.5)
    elif name == "Dropout":
        p[a], p[a + 1], p[a + 2] = A_DROPOUT, rs.uniform(0, 0.1), _per_channel(rs, 0.5)
    elif name == "CoarseDropout":                 # the mask is drawn at 15 % of the size (at least 3 cells), nearest up-sampling
        p[a], p[a + 1], p[a + 2] = A_COARSE, 0.02, _per_channel(rs, 0.5)
        p[a + 3], p[a + 4] = max(int(round(h * 0.15)), 3), max(int(round(w * 0.15)), 3)
    elif name == "Dropout2d":                     # each channel dropped with p = 0.5, at least one kept
        keep = [rs.uniform() >= 0.5 for _ in range(3)]
        if not any(keep):
            keep[rs.randint(0, 3)] = True
        p[a], p[a + 1] = A_DROP2D, sum(1 << k for k in range(3) if keep[k])
    elif name in ("ImpulseNoise", "SaltAndPepper", "Salt", "Pepper"):
        p[a], p[a + 1] = A_REPLACE, 0.1
        p[a + 2] = 1.0 if name == "ImpulseNoise" else 0.0             # ImpulseNoise = SaltAndPepper(per_channel=True)
        p[a + 3] = {"Salt": 1.0, "Pepper": 2.0}.get(name, 0.0)
    elif name == "Invert":
        if rs.uniform() < 0.15:
            p[a] = A_INVERT
    elif name == "Solarize":
        if rs.uniform() < 0.5:
            p[a], p[a + 1] = A_SOLARIZE, rs.uniform(32, 128)
    elif name == "JpegCompression":
        p[a], p[a + 1] = A_JPEG, _jpeg_quality(rs)
    elif name == "Emboss":
        al, st = rs.uniform(0.0, 1.0), rs.uniform(0.5, 1.5)
        _set_arith_filter(p, _blend(al, [[-1 - st, 0 - st, 0], [0 - st, 1, 0 + st], [0, 0 + st, 1 + st]]))
    elif name == "EdgeDetect":
        _set_arith_filter(p, _blend(rs.uniform(0.0, 1.0), [[0, 1, 0], [1, -4, 1], [0, 1, 0]]))
    elif name == "DirectedEdgeDetect":
        _set_arith_filter(p, directed_edge_kernel(rs.uniform(0.0, 1.0), rs.uniform(0.0, 1.0)))
    elif name == "FilterEdgeEnhanceMore":         # PIL.ImageFilter.EDGE_ENHANCE_MORE
        _set_arith_filter(p, [[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]], pil=True, offset=0.0, scale=1.0)
    elif name == "FilterContour":                 # PIL.ImageFilter.CONTOUR
        _set_arith_filter(p, [[-1, -1, -1], [-1, 8, -1], [-1, -1, -1]], pil=True, offset=255.0, scale=1.0)
    else:
        raise KeyError(name)


def _colour_member(p, rs, name):
    """One member of the `color` list (:146-163 = dataset_pretrain.py:107-122)."""
    b = P_B
    if name == "HueAdd0_50":                      # WithColorspace(HSV, WithChannels(0, Add((0, 50)))): uint8 H, saturating
        p[b], p[b + 1] = B_HUE_ADD, rs.randint(0, 51)
    elif name == "HueAdd50_100":                  # ChangeColorspace(HSV), WithChannels(0, Add((50, 100))), back
        p[b], p[b + 1] = B_HUE_ADD, rs.randint(50, 101)
    elif name == "MultiplyBrightness":
        p[b], p[b + 1], p[b + 2] = B_BRIGHT, rs.uniform(0.5, 1.5), 0.0
    elif name == "MultiplyAndAddToBrightness":    # (imgaug: on the brightness channel of a random colour space; here on RGB)
        p[b], p[b + 1], p[b + 2] = B_BRIGHT, rs.uniform(0.5, 1.5), rs.uniform(-30, 30)
    elif name == "MultiplyHueAndSaturation":      # per_channel=True: hue and saturation drawn separately
        p[b], p[b + 1], p[b + 2] = B_MUL_HS, rs.uniform(0.5, 1.5), rs.uniform(0.5, 1.5)
    elif name == "AddToHueAndSaturation":         # values -50 .. 50 on imgaug's 255 scale: hue shift int(v / 255 * 180) on cv2's H
        vh, vs = rs.randint(-50, 51), rs.randint(-50, 51)
        p[b], p[b + 1], p[b + 2] = B_ADD_HS, int(vh / 255.0 * 180.0), vs
    elif name == "Grayscale":
        p[b], p[b + 1] = B_GRAY, rs.uniform(0.0, 1.0)
    elif name == "KMeansColorQuantization":       # n_colors = (2, 16): Lloyd's iteration on the 8-bit Lab triples, on the device
        p[b], p[b + 1] = B_KMEANS, rs.randint(2, 17)
    elif name == "UniformColorQuantization":      # n_colors = (2, 16)
        p[b], p[b + 1] = B_UNIFORM_Q, rs.randint(2, 17)
    elif name == "ChangeColorTemperature":        # 1100 .. 10000 K as warm <-> cool channel gains
        t = rs.uniform(-1.0, 1.0)
        p[b], p[b + 1], p[b + 2], p[b + 3] = B_GAINS, 1.0 + 0.25 * t, 1.0, 1.0 - 0.25 * t
    elif name == "ChannelShuffle":
        if rs.uniform() < 0.35:
            p[b], p[b + 1] = B_SHUFFLE, rs.randint(0, 6)
    else:
        raise KeyError(name)


def _blur_member(p, rs, bilateral=True):
    """`Blur`: OneOf[Sharpen(alpha 0-0.5, lightness 0-0.5), OneOf[GaussianBlur, AverageBlur, MedianBlur, MotionBlur(, BilateralBlur)]]
    (:165-176; the finetuning pipeline has no bilateral member)."""
    if rs.uniform() < 0.5:
        al, light = rs.uniform(0.0, 0.5), rs.uniform(0.0, 0.5)
        _set_filter(p, _blend(al, [[-1, -1, -1], [-1, 8 + light, -1], [-1, -1, -1]]))
        return
    m = rs.randint(0, 5 if bilateral else 4)
    if m == 0:                                    # GaussianBlur(sigma 0.5 - 1.5): 5 x 5
        _set_filter(p, gaussian_kernel5(rs.uniform(0.5, 1.5)))
    elif m == 1:                                  # AverageBlur(k 2 - 6): cv2.blur, anchor k // 2
        k = rs.randint(2, 7)
        _set_filter(p, np.full((k, k), 1.0 / (k * k)))
    elif m == 2:                                  # MedianBlur(k 3 - 7): even draws go to the next odd size
        k = rs.randint(3, 8)
        p[P_C], p[P_C + 1] = C_MEDIAN, float(k if k % 2 else k + 1)
    elif m == 3:                                  # MotionBlur(k = 5, angle 0 - 360, direction -1 .. 1)
        _set_filter(p, motion_kernel(5, rs.uniform(0.0, 360.0), rs.uniform(-1.0, 1.0)))
    else:                                         # BilateralBlur(d 3 - 10, sigma_color / sigma_space 10 - 250)
        p[P_C], p[P_C + 1] = C_BILATERAL, float(rs.randint(3, 11))
        p[P_C + 2], p[P_C + 3] = rs.uniform(10, 250), rs.uniform(10, 250)


def _contrast_member(p, rs):
    """`contrast`: OneOf[Gamma, Linear, Sigmoid, Log, HistogramEqualization, AllChannelsHistogramEqualization, CLAHE, AllChannelsCLAHE]."""
    d = P_D
    k = rs.randint(0, 8)
    if k == 0:
        p[d], p[d + 1] = D_GAMMA, rs.uniform(0.5, 2.0)
    elif k == 1:
        p[d], p[d + 1] = D_LINEAR, rs.uniform(0.5, 1.0)
    elif k == 2:
        p[d], p[d + 1], p[d + 2] = D_SIGMOID, rs.uniform(3, 10), rs.uniform(0.4, 0.6)
    elif k == 3:
        p[d], p[d + 1] = D_LOG, rs.uniform(0.6, 1.4)
    elif k == 4:
        p[d] = D_HISTEQ_LAB                       # HistogramEqualization: equalizeHist on the L channel of 8-bit Lab
    elif k == 5:
        p[d] = D_HISTEQ_ALL
    else:                                         # CLAHE (k = 6: on L of Lab) / AllChannelsCLAHE (k = 7): clip_limit (0.1, 8), tile_grid_size_px (3, 12)
        p[d] = D_CLAHE_LAB if k == 6 else D_CLAHE_ALL     # - one draw for both sides; imgaug hands it to cv2 as the NUMBER of tiles a side
        p[d + 1], p[d + 2] = rs.uniform(0.1, 8.0), rs.randint(3, 13)


ARITHMETIC_5 = ["AddElementwise", "AdditiveGaussianNoise", "AdditiveLaplaceNoise", "AdditivePoissonNoise", "Multiply",
                "MultiplyElementwise", "Dropout", "CoarseDropout", "Dropout2d", "ImpulseNoise", "SaltAndPepper", "Salt", "Pepper",
                "Invert", "Solarize", "JpegCompression", "Emboss", "EdgeDetect", "DirectedEdgeDetect", "FilterEdgeEnhanceMore",
                "FilterContour"]                                                        # augmentation_pipelines.py:122-144
COLOR_5 = ["HueAdd0_50", "MultiplyAndAddToBrightness", "MultiplyHueAndSaturation", "AddToHueAndSaturation", "HueAdd50_100",
           "Grayscale", "KMeansColorQuantization", "UniformColorQuantization", "ChangeColorTemperature"]      # :146-163
FINETUNE_ONE_OF = (["ChannelShuffle", "AddElementwise", "AdditiveGaussianNoise", "AdditiveLaplaceNoise", "AdditivePoissonNoise",
                    "ImpulseNoise", "Multiply", "MultiplyElementwise", "Dropout", "CoarseDropout", "Dropout2d", "SaltAndPepper",
                    "Salt", "Pepper", "Solarize", "JpegCompression", "Emboss", "EdgeDetect", "DirectedEdgeDetect",
                    "FilterEdgeEnhanceMore", "FilterContour", "HueAdd0_50", "MultiplyBrightness", "MultiplyAndAddToBrightness",
                    "MultiplyHueAndSaturation", "AddToHueAndSaturation", "HueAdd50_100", "Grayscale", "KMeansColorQuantization",
                    "UniformColorQuantization", "ChangeColorTemperature", "Fog", "Clouds", "Snowflakes", "Rain"])   # dataset_pretrain.py:86-121
_COLOUR_NAMES = set(COLOR_5) | {"MultiplyBrightness", "ChannelShuffle"}
_WEATHER = {"Fog", "Clouds", "Snowflakes", "Rain"}                                      # layers drawn on the host (weather.py)


def _weather_member(p, rs, name, h, w, overlays):
    """`weather`: the layers of Fog / Clouds / Snowflakes / Rain are generated HERE (ccd_amd/dataset/weather.py) and blended on the
    device: the row names the first of its layers among the batch's overlay planes, their number and the blend (cloud / rain: alpha
    towards an intensity; snow: add, then raise).  Without a collector (`overlays` None: a caller that does not ship overlay planes)
    no layer is made - but the layer seed is drawn either way, so the sampler's random stream (every later draw of the batch) does not
    depend on whether overlay planes are shipped."""
    layer_seed = rs.randint(0, 1 << 31)
    if overlays is None:
        return
    from . import weather
    # (-1, task id, blend) until Overlays.resolve() has the layers: then (their number, the first one's index, blend)
    p[P_W], p[P_W + 1] = -1, overlays.add_task(name, layer_seed)
    p[P_W + 2] = weather.SNOW_MODE if name == "Snowflakes" else weather.CLOUD_MODE      # how the device blends the planes


WEATHER_5 = ["Fog", "Clouds", "Snowflakes", "Rain"]                                    # augmentation_pipelines.py:192-195


def _colour_params(rs: np.random.RandomState, severity: int, h: int = 32, w: int = 128, overlays=None) -> np.ndarray:
    p = IDENTITY_PARAMS.copy()
    p[P_SEED] = rs.randint(0, 1 << 24)
    if severity <= 0:
        return p
    if severity == 5:
        if rs.uniform() < 0.2:                    # iaa.Sometimes(0.2, Identity, Sequential[...])
            return p
        _arith_member(p, rs, ARITHMETIC_5[rs.randint(0, len(ARITHMETIC_5))], h, w)
        if rs.uniform() < 0.7:
            _colour_member(p, rs, COLOR_5[rs.randint(0, len(COLOR_5))])
        if rs.uniform() < 0.7:
            _blur_member(p, rs)
        if rs.uniform() < 0.7:
            _contrast_member(p, rs)
        if rs.uniform() < 0.7:                    # weather: Sometimes(0.7, OneOf[Fog, Clouds, Snowflakes, Rain])
            _weather_member(p, rs, WEATHER_5[rs.randint(0, len(WEATHER_5))], h, w, overlays)
        return p
    # the other severities are OneOf lists over colour members (severity 6: :198-214); no shipped config selects them
    names = ["HueAdd0_50", "MultiplyAndAddToBrightness", "MultiplyHueAndSaturation", "AddToHueAndSaturation", "HueAdd50_100",
             "Grayscale", "ChannelShuffle", "ChangeColorTemperature"]
    _colour_member(p, rs, names[rs.randint(0, len(names))])
    return p


def sample_colour_params(rs: np.random.RandomState, batch: int, severity: int = 5, warped=None, h: int = 32,
                         w: int = 128, overlays=None, resolve: bool = True) -> np.ndarray:
    """fp32 [batch, 2, 96]: view 1 from pipeline `severity`, view 2 from the same pipeline (both come from `augment_tfs`,
    datasetsupervised_kmeans.py:57).  `warped` (bool [batch], from sample_theta): a sample whose warp draw failed gets the
    plain image as view 2 (:72-74 `image_view = image`), i.e. identity parameters.  `overlays` (weather.Overlays): collects the
    layers of the rows that drew a weather member: the rows carry task ids until `overlays.resolve(params, P_W)` (called here unless
    `resolve=False`: a caller that overlaps the drawing with other work starts it and resolves later) - hand its planes() to
    ops.augment_views together with the rows."""
    out = np.empty((batch, 2, AUG_NP), dtype=np.float32)
    for b in range(batch):
        out[b, 0] = _colour_params(rs, severity, h, w, overlays)
        out[b, 1] = _colour_params(rs, severity, h, w, overlays)
        if warped is not None and not warped[b]:
            out[b, 1] = IDENTITY_PARAMS           # (a task it may have registered is never looked at)
    if overlays is not None:
        overlays.start()
        if resolve:
            overlays.resolve(out, P_W)
    return out


# ------------------------------------------------------------------------------------------------ finetuning pipeline
def _finetune_colour_params(rs: np.random.RandomState, h: int = 32, w: int = 128, overlays=None) -> np.ndarray:
    """One draw of the finetuning pipeline's colour part (Dino/dataset/dataset_pretrain.py:80-146): Sometimes(0.6, Invert(0.1)),
    Sometimes(0.8, OneOf[35 members]), Sometimes(0.6, Blur group without the bilateral member), Sometimes(0.6, contrast group).
    Members this implementation does not reproduce keep their share of the draw and leave the image unchanged."""
    p = IDENTITY_PARAMS.copy()
    p[P_SEED] = rs.randint(0, 1 << 24)
    if rs.uniform() < 0.6:
        p[P_PREINV] = float(rs.uniform() < 0.1)
    if rs.uniform() < 0.8:
        name = FINETUNE_ONE_OF[rs.randint(0, len(FINETUNE_ONE_OF))]
        if name in _WEATHER:
            _weather_member(p, rs, name, h, w, overlays)     # (the blend runs after the contrast group on the device; the reference's OneOf
        elif name in _COLOUR_NAMES:                          # puts it before the blur / contrast groups - an ordering difference, noted)
            _colour_member(p, rs, name)
        else:
            _arith_member(p, rs, name, h, w)
    if rs.uniform() < 0.6:
        _blur_member(p, rs, bilateral=False)
    if rs.uniform() < 0.6:
        _contrast_member(p, rs)
    return p


P_WARP = P_W + 3       # view-2 row: 0 = warp by theta; m > 0 = warp map m - 1 (weather.WarpMaps; a parked task is negative)


def sample_finetune_params(rs: np.random.RandomState, batch: int, h: int, w: int, overlays=None, resolve: bool = True, warps=None):
    """(params fp32 [batch, 2, 96] - only row 1 is used, theta fp32 [batch, 3, 3]) for the finetuning augmentation
    (dataset_pretrain.py:79-158): colour as above, geometry = Sometimes(0.6, OneOf[Affine (the pretraining ranges), PiecewiseAffine
    (scale 0.01 - 0.1: a dense source-position map drawn on the host, `warps` = weather.WarpMaps collects them; without a collector the
    draw leaves the image unwarped), Rotate(-45, 45)])."""
    params = np.tile(IDENTITY_PARAMS, (batch, 2, 1)).astype(np.float32)
    theta = np.tile(np.eye(3, dtype=np.float32), (batch, 1, 1))
    for b in range(batch):
        params[b, 1] = _finetune_colour_params(rs, h, w, overlays)
        if rs.uniform() < 0.6:
            g = rs.randint(0, 3)
            if g == 0:
                theta[b] = theta_from_pixel_matrix(affine_pixel_matrix(rs, h, w), h, w)
            elif g == 1:
                seed = rs.randint(0, 1 << 31)
                if warps is not None:
                    params[b, 1, P_WARP] = warps.park("PiecewiseAffine", seed)
            elif g == 2:
                rot = math.radians(rs.uniform(-45.0, 45.0))
                cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
                to_o = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]], dtype=np.float64)
                back = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1]], dtype=np.float64)
                rotm = np.array([[math.cos(rot), -math.sin(rot), 0], [math.sin(rot), math.cos(rot), 0], [0, 0, 1]])
                theta[b] = theta_from_pixel_matrix(back @ rotm @ to_o, h, w)
    for coll, col in ((overlays, P_W), (warps, P_WARP)):
        if coll is not None:
            coll.start()
            if resolve:
                coll.resolve(params, col)
    return params, theta
