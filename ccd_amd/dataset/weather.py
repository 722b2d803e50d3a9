"""The reference's `weather` group (augmentation_pipelines.py:187-196: iaa.Fog, iaa.Clouds, iaa.Snowflakes, iaa.Rain; the finetuning
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


# ---------------------------------------------------------------------------------------------------- Snowflakes / Rain
# imgaug's SnowflakesLayer / RainLayer (augmenters/weather.py): salt noise on a canvas shrunk by the flake size, thinned by a coarse
# Beta-distributed gate, up-sampled (cubic), Gaussian-blurred a little (snow only) and smeared by a motion-blur kernel along the
# falling direction.  Snow is blended by sum (a faint glow) then by maximum (the flakes); rain is an alpha blend towards a grey drop
# colour - the same form as a cloud layer, alpha = noise / 255.
SNOW_MODE, CLOUD_MODE = 1, 0


def _salt_canvas(rs, h, w, density):
    """arithmetic.Salt(p=density) on a black uint8 canvas: a pixel is replaced with probability `density` by the upper half of
    255 * Beta(0.5, 0.5)."""
    hit = rs.rand(h, w) < density
    val = 0.5 + np.abs(rs.beta(0.5, 0.5, size=(h, w)) - 0.5)
    return np.where(hit, np.clip(np.round(val * 255.0), 0, 255), 0).astype(np.uint8)


def _falling_noise(rs, h, w, *, density, density_uniformity, flake_size, flake_size_uniformity, angle, speed, blur_sigma_fraction,
                   blur: bool):
    """-> (noise uint8 [h, w] after the blurs, speed, flake_size_uniformity) of one layer."""
    from scipy import ndimage
    from .augment import motion_kernel
    flake = _draw(rs, flake_size)
    uniformity = _draw(rs, flake_size_uniformity)
    ang, spd = _draw(rs, angle), _draw(rs, speed)
    sig_frac = _draw(rs, blur_sigma_fraction)
    dens, dens_uni = _draw(rs, density), _draw(rs, density_uniformity)
    down = min(max(1.0 - flake, 0.001), 1.0)
    hd, wd = max(1, int(h * down)), max(1, int(w * down))
    noise = _salt_canvas(rs, hd, wd, dens)
    gate = resize_cubic(rs.beta(1.0, max(1.0 - dens_uni, 1e-6), size=(8, 8)), hd, wd)          # most of its weight near 1: little gating
    noise = np.clip(noise.astype(np.float64) * np.clip(gate, 0.0, 1.0), 0, 255).astype(np.uint8)
    noise = resize_cubic(noise, h, w, as_uint8=True)
    if blur:
        sigma = min(max(max(h, w) * sig_frac, 0.5), 3.75)
        noise = np.clip(np.round(ndimage.gaussian_filter(noise, sigma=sigma, mode="mirror")), 0, 255)
    k = int(spd * max(h, w))
    if k > 1:
        kern = motion_kernel(max(k, 3), ang, 1.0)
        noise = np.clip(np.round(ndimage.correlate(noise.astype(np.float64), kern, mode="mirror")), 0, 255)
    return noise.astype(np.uint8), spd, uniformity


SNOWFLAKES = dict(density=(0.005, 0.075), density_uniformity=(0.3, 0.9), flake_size=(0.1, 0.4), flake_size_uniformity=(0.4, 0.8),
                  angle=(-30, 30), speed=(0.01, 0.05), blur_sigma_fraction=(0.0001, 0.001))      # iaa.Snowflakes(flake_size=(0.1, 0.4), speed=(0.01, 0.05))
RAIN = dict(density=(0.03, 0.14), density_uniformity=(0.8, 1.0), flake_size=(0.01, 0.02), flake_size_uniformity=(0.2, 0.5),
            angle=(-15, 15), speed=(0.1, 0.3), blur_sigma_fraction=(0.001, 0.001))                 # iaa.Rain(speed=(0.1, 0.3))


def snowflake_layers(rs, h, w):
    """iaa.Snowflakes: SomeOf((1, 3)) of three identically-specified layers.  Planes of a layer: (what is ADDED, what the result is
    raised to at least) - out = max(clip(v + plane0, 0, 255), plane1), the device's snow mode."""
    out = []
    for _ in range(rs.randint(1, 4)):
        noise, spd, uni = _falling_noise(rs, h, w, blur=True, **SNOWFLAKES)
        gain, gain_adj = 1.0 + 2.0 * (1.0 - uni), 1.0 + 5.0 * (1.0 - uni)
        n = np.floor(255.0 * np.power(noise.astype(np.float64) / 255.0, gain)) * gain_adj        # GammaContrast's truncating table, then the re-gain
        out.append(np.stack([(0.1 + 20.0 * spd) * n, (1.0 + 20.0 * spd) * n]).astype(np.float32))
    return out


def rain_layers(rs, h, w):
    """iaa.Rain: 1 - 3 RainLayers; a layer is an alpha blend towards the drop colour the library derives from the noise itself
    (110 + (240 - 110) % sum of its first 1000 values): planes (alpha = noise / 255, intensity = that colour) - the cloud blend."""
    out = []
    for _ in range(rs.randint(1, 4)):
        noise, _, _ = _falling_noise(rs, h, w, blur=False, **RAIN)
        total = float(noise.reshape(-1)[:1000].astype(np.float64).sum() * 3.0)                    # (the library sums the RGB-tiled noise)
        colour = 110.0 + (130.0 % (total if total > 0 else 1.0))
        out.append(np.stack([noise.astype(np.float32) / 255.0, np.full((h, w), colour, np.float32)]).astype(np.float32))
    return out


def piecewise_affine_map(rs, h, w, nb_rows=4, nb_cols=4, scale=(0.01, 0.1)):
    """iaa.PiecewiseAffine(scale=(0.01, 0.1)) (the finetuning geometry's second member, dataset_pretrain.py:156) as a dense map:
    -> [fp32 [2, h, w]] = the (x, y) SOURCE position of every output pixel.  imgaug: a regular nb_rows x nb_cols mesh over
    [0, h] x [0, w], every point moved by Normal(0, s) * (h, w) with ONE s ~ U(scale) per image and clipped into the image;
    skimage.transform.PiecewiseAffineTransform.estimate(mesh, moved) = scipy's Delaunay triangulation of the REGULAR mesh + the affine
    map of each triangle onto its moved copy, used by skimage.transform.warp as the inverse map (output pixel -> source position;
    -1 outside the mesh, which covers every pixel here).  The triangulation is scipy's own (what skimage calls): the diagonal
    qhull picks in each square cell of the degenerate mesh is the library's, not restated."""
    from scipy.spatial import Delaunay
    s = rs.uniform(*scale)
    jitter = rs.normal(0.0, s, size=(nb_rows * nb_cols, 2))
    xx, yy = np.meshgrid(np.linspace(0, w, nb_cols), np.linspace(0, h, nb_rows))
    src = np.stack([xx.ravel(), yy.ravel()], 1)                                     # (x, y)
    dst = src + jitter[:, ::-1] * np.array([w, h], np.float64)                      # (jitter columns are (y, x) in the library)
    dst[:, 0], dst[:, 1] = np.clip(dst[:, 0], 0, w - 1), np.clip(dst[:, 1], 0, h - 1)
    tri = Delaunay(src)
    py, px = np.mgrid[0:h, 0:w]
    pts = np.stack([px.ravel(), py.ravel()], 1).astype(np.float64)
    simplex = tri.find_simplex(pts)
    out = np.full((h * w, 2), -1.0)
    ones = np.ones((3, 1))
    for k, verts in enumerate(tri.simplices):
        sel = simplex == k
        if sel.any():
            a = np.linalg.solve(np.hstack([src[verts], ones]), dst[verts])          # [x y 1] a = [x' y']
            out[sel] = np.hstack([pts[sel], np.ones((int(sel.sum()), 1))]) @ a
    return [np.ascontiguousarray(out.T.reshape(2, h, w), dtype=np.float32)]


# member -> (generator of a list of [2, h, w] planes, the dtype they travel in)
MAKERS = {"Fog": (fog_layers, np.float16), "Clouds": (clouds_layers, np.float16), "Snowflakes": (snowflake_layers, np.float16),
          "Rain": (rain_layers, np.float16), "PiecewiseAffine": (piecewise_affine_map, np.float32)}


def draw_layers(task):
    """(member name, seed, h, w) -> the member's planes [n, 2, h, w] (fp16 weather layers, fp32 warp maps): a pure function of its
    arguments (worker processes call it)."""
    name, seed, h, w = task
    make, dtype = MAKERS[name]
    return np.ascontiguousarray(np.stack(make(np.random.RandomState(int(seed)), int(h), int(w))), dtype=dtype)


class LayerFarm:
    """A pool of worker processes that draw weather layers (~1 ms of numpy / scipy each; a batch of 256 samples needs ~500 of them:
    1.1 s on one core against a 50-ms training step).  Forked lazily, like a DataLoader's workers; the workers never touch the GPU."""

    def __init__(self, workers=None):
        import os
        # (default: half the cores, at most 32 - shared between the ranks of the node: 8 ranks x 32 workers beside 8 x the loader's own
        # workers would oversubscribe a 256-core host)
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
        self.workers = int(workers) if workers is not None else max(1, min(32, (os.cpu_count() or 2) // (2 * ranks)))
        self.pool = None

    def submit(self, tasks):
        """-> an object with .get() -> [fp16 [n, 2, h, w]] in task order."""
        if not tasks:
            return _Ready([])
        if self.workers <= 1:
            return _Ready([draw_layers(t) for t in tasks])
        if self.pool is None:
            import multiprocessing as mp
            self.pool = mp.get_context("fork").Pool(self.workers)
        return self.pool.map_async(draw_layers, tasks, chunksize=max(1, len(tasks) // (4 * self.workers)))

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Ready:
    def __init__(self, value):
        self.value = value

    def get(self, timeout=None):
        return self.value


class Overlays:
    """The weather layers of one batch.  The sampler registers a task per (sample, view) row that drew a weather member
    (`add_task` -> a task id, parked in the row); `start()` hands the tasks to a LayerFarm (or draws them in place) and `resolve(params)`
    waits, stacks the layers and writes every row's first layer and layer count: `planes()` -> fp16 [layers, 2, H, W] (None when nobody
    drew one).  A layer depends on (member, seed) only, so a batch is reproducible whatever the workers' scheduling."""

    def __init__(self, h: int, w: int, farm: "LayerFarm | None" = None):
        self.h, self.w, self.farm = int(h), int(w), farm
        self.tasks, self.pending, self._planes = [], None, None

    def add_task(self, name: str, seed: int) -> int:
        self.tasks.append((name, int(seed), self.h, self.w))
        return len(self.tasks) - 1

    def start(self):
        if self.pending is None:
            self.pending = self.farm.submit(self.tasks) if self.farm is not None else _Ready([draw_layers(t) for t in self.tasks])
        return self

    def resolve(self, params: np.ndarray, p_w: int):
        """params: fp32 [..., AUG_NP] whose weather rows carry (-1, task id, blend) at p_w .. p_w + 2 -> (count, first layer, blend)."""
        self.start()
        try:
            results = self.pending.get(timeout=120.0)
        except Exception as exc:                         # a wedged or killed worker must not stall the training: draw here, say so once
            import warnings
            warnings.warn(f"weather-layer workers did not answer ({type(exc).__name__}); drawing {len(self.tasks)} layers in the training process")
            results = [draw_layers(t) for t in self.tasks]
        flat = params.reshape(-1, params.shape[-1])
        rows = np.nonzero(flat[:, p_w] < 0)[0]
        layers, first = [], {}
        for r in rows:                                   # (rows that were overwritten - an unwarped view 2 - no longer refer to their task)
            tid = int(flat[r, p_w + 1])
            if tid not in first:
                first[tid] = sum(len(x) for x in layers)
                layers.append(results[tid])
            flat[r, p_w], flat[r, p_w + 1] = len(results[tid]), first[tid]
        self._planes = np.concatenate(layers) if layers else None
        return self._planes

    def planes(self):
        return self._planes


class WarpMaps(Overlays):
    """The piecewise-affine warp maps of one batch (fp32 [maps, 2, H, W], ops.augment_views' `warp_maps`): the same hand-over as the
    weather layers - a row parks -(task id + 1) in ONE column, `resolve` turns it into map index + 1 (0 = the row warps by theta)."""

    def park(self, name: str, seed: int) -> float:
        return -float(self.add_task(name, seed) + 1)

    def resolve(self, params: np.ndarray, col: int):
        self.start()
        try:
            results = self.pending.get(timeout=120.0)
        except Exception as exc:
            import warnings
            warnings.warn(f"warp-map workers did not answer ({type(exc).__name__}); drawing {len(self.tasks)} maps in the training process")
            results = [draw_layers(t) for t in self.tasks]
        flat = params.reshape(-1, params.shape[-1])
        maps = []
        for r in np.nonzero(flat[:, col] < 0)[0]:
            maps.append(results[int(-flat[r, col]) - 1])
            flat[r, col] = len(maps)
        self._planes = np.concatenate(maps) if maps else None
        return self._planes
