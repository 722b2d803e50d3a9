"""The cloud-layer members of the reference's `weather` group (augmentation_pipelines.py:187-196: iaa.Fog, iaa.Clouds; the finetuning
list dataset_pretrain.py:117-120 has them too) - host side.

imgaug draws a cloud layer as two low-frequency noise maps per image, an opacity `alpha` and an `intensity`, and blends
    out = clip((1 - alpha) * image + alpha * intensity, 0, 255)              (uint8, truncated)
The maps are frequency noise (`iap.FrequencyNoise`: white noise with random phases shaped by |f|^exponent, inverse FFT, normalised
to 0..1, generated at no more than `size_px_max` pixels on the longer side and up-sampled with cv2's bicubic kernel through an 8-bit
image) - a few hundred microseconds of numpy per (sample, view) that gets one, in the sampler that already draws every other
augmentation parameter on the host.  The maps travel to the device as fp16 planes `[layers, 2, H, W]` (alpha, intensity) and
`augment_spatial_kernel` (csrc/kernels/datapipe.h) applies the blend after the `contrast` group; a (sample, view)'s parameter row
names its first layer and their number (Fog: one layer, Clouds: one or two of its two layer kinds, in order).

Restated from imgaug 0.4.0's published algorithm (imgaug.parameters.FrequencyNoise, imgaug.augmenters.weather.CloudLayer / Fog /
Clouds); imgaug and cv2 are not part of this image, so the restatement is UNPINNED against the libraries: what the tests check
is the blend (device vs numpy, value by value) and the statistics of the maps.  The random stream is the sampler's numpy
RandomState, not imgaug's: distributions, not draws, are reproduced - as for every other member.
"""
from __future__ import annotations

import numpy as np


def _cubic_coeffs(x):
    """cv2's bicubic taps (A = -0.75) for the fractional offsets x: [..., 4]."""
    a = -0.75
    x = np.asarray(x, dtype=np.float64)
    c0 = ((a * (x + 1) - 5 * a) * (x + 1) + 8 * a) * (x + 1) - 4 * a
    c1 = ((a + 2) * x - (a + 3)) * x * x + 1
    c2 = ((a + 2) * (1 - x) - (a + 3)) * (1 - x) * (1 - x) + 1
    return np.stack([c0, c1, c2, 1.0 - c0 - c1 - c2], axis=-1)


def resize_cubic(src: np.ndarray, h: int, w: int, as_uint8: bool = False) -> np.ndarray:
    """cv2.resize(src, (w, h), interpolation=INTER_CUBIC) of a 2-D float array: half-pixel centres, replicated border.  as_uint8: the
    8-bit variant's rounding and saturation at the end (its fixed-point taps differ from this in the last level at most)."""
    src = np.asarray(src, dtype=np.float64)
    sh, sw = src.shape

    def axis(n_src, n_dst):
        f = (np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(np.int64)
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, _cubic_coeffs(f - i0)

    iy, cy = axis(sh, h)
    ix, cx = axis(sw, w)
    rows = (src[:, ix] * cx[None, :, :]).sum(-1)                      # [sh, w]
    out = (rows[iy, :] * cy[:, :, None]).sum(1)                       # [h, w]
    if as_uint8:
        out = np.clip(np.floor(out + 0.5), 0, 255)
    return out


def frequency_noise(rs: np.random.RandomState, h: int, w: int, exponent: float, size_px_max: float) -> np.ndarray:
    """iap.FrequencyNoise(exponent, size_px_max, upscale_method="cubic").draw_samples((h, w)) -> float32 [h, w] in 0..1."""
    maxlen = max(h, w)
    if maxlen > size_px_max:
        scale = size_px_max / maxlen
        hs, ws = int(h * scale), int(w * scale)
    else:
        hs, ws = h, w
    hs, ws = max(hs, 4), max(ws, 4)                                   # (the library does not go below 4 pixels a side)
    wn_r = rs.rand(hs, ws) * (max(hs, ws) ** 2)
    wn_a = rs.rand(hs, ws) * 2.0 * np.pi
    wn_r = wn_r * np.cos(wn_a)
    wn_i = wn_r * np.sin(wn_a)                                        # (sic: the library multiplies the already-rotated real part)
    yy, xx = np.mgrid[0:hs, 0:ws]
    f = np.sqrt(np.minimum(yy, hs - yy) ** 2.0 + np.minimum(xx, ws - xx) ** 2.0)
    f[0, 0] = 1.0
    shape = f ** exponent
    shape[0, 0] = 0.0
    inv = np.fft.ifft2((wn_r + 1j * wn_i) * shape).real
    lo, hi = inv.min(), inv.max()
    noise = (inv - lo) / (hi - lo) if hi > lo else np.zeros_like(inv)
    if (hs, ws) != (h, w):
        noise = resize_cubic(np.floor(noise * 255.0), h, w, as_uint8=True) / 255.0       # through an 8-bit image, as the library does
    return noise.astype(np.float32)


def _draw(rs, v):
    return float(rs.uniform(v[0], v[1])) if isinstance(v, tuple) else float(v)


def cloud_layer(rs: np.random.RandomState, h: int, w: int, *, intensity_mean, intensity_freq_exponent, intensity_coarse_scale,
                alpha_min, alpha_multiplier, alpha_size_px_max, alpha_freq_exponent, sparsity, density_multiplier) -> np.ndarray:
    """CloudLayer.generate_maps -> float32 [2, h, w]: (alpha in 0..1, intensity in 0..255)."""
    mean = _draw(rs, intensity_mean)
    a_min, a_mul = _draw(rs, alpha_min), _draw(rs, alpha_multiplier)
    a_px = _draw(rs, alpha_size_px_max)
    i_exp, a_exp = _draw(rs, intensity_freq_exponent), _draw(rs, alpha_freq_exponent)
    spars, dens = _draw(rs, sparsity), _draw(rs, density_multiplier)
    coarse = resize_cubic(mean + rs.normal(0.0, intensity_coarse_scale, size=(8, 8)), h, w)
    fine = mean * ((2.0 * frequency_noise(rs, h, w, i_exp, max(h, w, 1)) - 1.0) / 5.0)
    intensity = np.clip(coarse + fine, 0.0, 255.0)
    alpha = a_min + a_mul * frequency_noise(rs, h, w, a_exp, a_px)
    alpha = np.clip((alpha ** spars) * dens, 0.0, 1.0)
    return np.stack([alpha, intensity]).astype(np.float32)


FOG = dict(intensity_mean=(220, 255), intensity_freq_exponent=(-2.0, -1.5), intensity_coarse_scale=2, alpha_min=(0.7, 0.9),
           alpha_multiplier=0.3, alpha_size_px_max=(2, 8), alpha_freq_exponent=(-4.0, -2.0), sparsity=0.9,
           density_multiplier=(0.4, 0.9))                                                          # iaa.Fog
CLOUDS = (dict(intensity_mean=(196, 255), intensity_freq_exponent=(-2.5, -2.0), intensity_coarse_scale=10, alpha_min=0,
               alpha_multiplier=(0.25, 0.75), alpha_size_px_max=(2, 8), alpha_freq_exponent=(-2.5, -2.0), sparsity=(0.8, 1.0),
               density_multiplier=(0.5, 1.0)),
          dict(intensity_mean=(196, 255), intensity_freq_exponent=(-2.0, -1.0), intensity_coarse_scale=10, alpha_min=0,
               alpha_multiplier=(0.5, 1.0), alpha_size_px_max=(64, 128), alpha_freq_exponent=(-2.0, -1.0), sparsity=(1.0, 1.4),
               density_multiplier=(0.8, 1.5)))                                                     # iaa.Clouds: SomeOf((1, 2), these, in order)


def fog_layers(rs, h, w):
    return [cloud_layer(rs, h, w, **FOG)]


def clouds_layers(rs, h, w):
    n = rs.randint(1, 3)                                              # SomeOf((1, 2), ..., random_order=False)
    picked = sorted(rs.choice(2, size=n, replace=False).tolist())
    return [cloud_layer(rs, h, w, **CLOUDS[k]) for k in picked]


class Overlays:
    """Collects the layers a batch's parameter rows refer to: `planes()` -> fp16 [layers, 2, H, W] (or None when nobody drew one)."""

    def __init__(self, h: int, w: int):
        self.h, self.w, self.layers = h, w, []

    def add(self, layers) -> int:
        first = len(self.layers)
        self.layers.extend(layers)
        return first

    def planes(self):
        if not self.layers:
            return None
        return np.stack(self.layers).astype(np.float16)
