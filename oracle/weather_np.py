"""CPU restatement (numpy only) of the host-drawn augmentation layers - TEST INFRASTRUCTURE ONLY; the product never imports it.

What the product draws on the host (ccd_amd/dataset/weather.py) - imgaug's frequency noise, the cloud / fog layers, the snowflake and
rain layers (augmentation_pipelines.py:187-196, dataset_pretrain.py:117-120) and the dense source-position map of
iaa.PiecewiseAffine (dataset_pretrain.py:156) - restated HERE a second time from the published algorithms (imgaug 0.4.0:
parameters.FrequencyNoise, augmenters.weather.CloudLayer / SnowflakesLayer / RainLayer, augmenters.geometric.PiecewiseAffine;
skimage.transform.PiecewiseAffineTransform; cv2's bicubic resize and Gaussian blur; the mirror-mode correlation imgaug's motion blur
runs), in other
formulations than the product's: dense resampling matrices instead of tap gathers, `fftfreq` distances and one inverse FFT per axis,
explicit mirror padding and window sums instead of scipy.ndimage, barycentric interpolation instead of one linear solve per triangle.
tests/test_datapipe_cpu.py compares the two value by value on shared seeds, so the product's generators are no longer their own checker.

Shared with the product by construction (and therefore not independent): the ORDER in which a layer takes its draws from the numpy
RandomState (the product's stream is the specification there - imgaug's own generator is not reproduced by either side), and the
triangulation of the regular mesh, which both take from scipy.spatial.Delaunay - the call skimage itself makes.
PARITY UNPINNED against imgaug / cv2 / skimage: none of them is part of this image.
"""
import numpy as np


# ------------------------------------------------------------------------------------------------ cv2.resize, INTER_CUBIC
def keys_weight(d, a=-0.75):
    """Keys' cubic convolution kernel at distance d (cv2 uses a = -0.75)."""
    d = abs(float(d))
    if d <= 1.0:
        return (a + 2.0) * d ** 3 - (a + 3.0) * d ** 2 + 1.0
    if d < 2.0:
        return a * d ** 3 - 5.0 * a * d ** 2 + 8.0 * a * d - 4.0 * a
    return 0.0


def resample_matrix(n_src, n_dst):
    """[n_dst, n_src]: row i holds the four cubic taps of output sample i (half-pixel centres, border replicated: taps that fall
    outside are added onto the edge sample)."""
    m = np.zeros((n_dst, n_src), np.float64)
    for i in range(n_dst):
        pos = (i + 0.5) * n_src / n_dst - 0.5
        base = int(np.floor(pos))
        for tap in range(base - 1, base + 3):
            m[i, min(max(tap, 0), n_src - 1)] += keys_weight(pos - tap)
    return m


def resize_cubic(src, h, w, to_u8=False):
    src = np.asarray(src, np.float64)
    out = resample_matrix(src.shape[0], h) @ src @ resample_matrix(src.shape[1], w).T
    if to_u8:                                                  # cv2's 8-bit path: round half up, saturate
        out = np.minimum(np.maximum(np.floor(out + 0.5), 0.0), 255.0)
    return out


# ------------------------------------------------------------------------------------------------ iap.FrequencyNoise
def frequency_noise(rs, h, w, exponent, size_px_max):
    """White noise (uniform modulus, uniform phase) shaped by |f|^exponent, inverse transform, min-max normalised; drawn at no more than
    size_px_max pixels on the longer side (not below 4) and brought to (h, w) through an 8-bit image with the cubic resize."""
    longer = max(h, w)
    hs, ws = (int(h * (size_px_max / longer)), int(w * (size_px_max / longer))) if longer > size_px_max else (h, w)
    hs, ws = max(hs, 4), max(ws, 4)
    modulus = rs.rand(hs, ws) * float(max(hs, ws)) ** 2
    phase = rs.rand(hs, ws) * (2.0 * np.pi)
    re = modulus * np.cos(phase)
    im = re * np.sin(phase)                                    # the library's own formula: the imaginary part is built from `re`
    # distance of every bin from the zero frequency, in bins: |fftfreq| * n is min(k, n - k)
    fy, fx = np.abs(np.fft.fftfreq(hs)) * hs, np.abs(np.fft.fftfreq(ws)) * ws
    dist = np.hypot(fy[:, None], fx[None, :])
    gain = np.zeros_like(dist)
    nz = dist > 0
    gain[nz] = dist[nz] ** exponent                            # the zero frequency is dropped (no mean)
    spec = (re + 1j * im) * gain
    field = np.fft.ifft(np.fft.ifft(spec, axis=0), axis=1).real
    span = field.max() - field.min()
    noise = (field - field.min()) / span if span > 0 else np.zeros_like(field)
    if (hs, ws) != (h, w):
        noise = resize_cubic(np.floor(noise * 255.0), h, w, to_u8=True) / 255.0
    return noise.astype(np.float32)


def _param(rs, v):
    return float(rs.uniform(v[0], v[1])) if isinstance(v, tuple) else float(v)


# ------------------------------------------------------------------------------------------------ CloudLayer (Fog / Clouds)
FOG = dict(intensity_mean=(220, 255), intensity_freq_exponent=(-2.0, -1.5), intensity_coarse_scale=2, alpha_min=(0.7, 0.9),
           alpha_multiplier=0.3, alpha_size_px_max=(2, 8), alpha_freq_exponent=(-4.0, -2.0), sparsity=0.9, density_multiplier=(0.4, 0.9))
CLOUDS = [dict(intensity_mean=(196, 255), intensity_freq_exponent=(-2.5, -2.0), intensity_coarse_scale=10, alpha_min=0,
               alpha_multiplier=(0.25, 0.75), alpha_size_px_max=(2, 8), alpha_freq_exponent=(-2.5, -2.0), sparsity=(0.8, 1.0),
               density_multiplier=(0.5, 1.0)),
          dict(intensity_mean=(196, 255), intensity_freq_exponent=(-2.0, -1.0), intensity_coarse_scale=10, alpha_min=0,
               alpha_multiplier=(0.5, 1.0), alpha_size_px_max=(64, 128), alpha_freq_exponent=(-2.0, -1.0), sparsity=(1.0, 1.4),
               density_multiplier=(0.8, 1.5))]


def cloud_layer(rs, h, w, spec):
    """-> (alpha [h, w] in 0..1, intensity [h, w] in 0..255).  intensity = a coarse 8 x 8 normal field around the mean, up-sampled,
    + a fine frequency-noise field of +- mean / 5; alpha = (alpha_min + mul * noise) ^ sparsity * density, clipped."""
    mean = _param(rs, spec["intensity_mean"])
    a_min = _param(rs, spec["alpha_min"])
    a_mul = _param(rs, spec["alpha_multiplier"])
    a_px = _param(rs, spec["alpha_size_px_max"])
    i_exp = _param(rs, spec["intensity_freq_exponent"])
    a_exp = _param(rs, spec["alpha_freq_exponent"])
    sparsity = _param(rs, spec["sparsity"])
    density = _param(rs, spec["density_multiplier"])
    coarse = resize_cubic(rs.normal(0.0, spec["intensity_coarse_scale"], size=(8, 8)) + mean, h, w)
    fine = frequency_noise(rs, h, w, i_exp, max(h, w, 1)).astype(np.float64)
    intensity = np.minimum(np.maximum(coarse + mean * (2.0 * fine - 1.0) / 5.0, 0.0), 255.0)
    alpha = a_min + a_mul * frequency_noise(rs, h, w, a_exp, a_px).astype(np.float64)
    alpha = np.minimum(np.maximum(np.power(alpha, sparsity) * density, 0.0), 1.0)
    return alpha.astype(np.float32), intensity.astype(np.float32)


def fog_layers(rs, h, w):
    return [cloud_layer(rs, h, w, FOG)]


def clouds_layers(rs, h, w):
    count = rs.randint(1, 3)                                   # SomeOf((1, 2)) of the two layer kinds, kept in their listed order
    kinds = sorted(int(k) for k in rs.choice(2, size=count, replace=False))
    return [cloud_layer(rs, h, w, CLOUDS[k]) for k in kinds]


# ------------------------------------------------------------------------------------------------ SnowflakesLayer / RainLayer
SNOWFLAKES = dict(density=(0.005, 0.075), density_uniformity=(0.3, 0.9), flake_size=(0.1, 0.4), flake_size_uniformity=(0.4, 0.8),
                  angle=(-30, 30), speed=(0.01, 0.05), blur_sigma_fraction=(0.0001, 0.001))
RAIN = dict(density=(0.03, 0.14), density_uniformity=(0.8, 1.0), flake_size=(0.01, 0.02), flake_size_uniformity=(0.2, 0.5),
            angle=(-15, 15), speed=(0.1, 0.3), blur_sigma_fraction=(0.001, 0.001))


def mirror_pad(a, r, axis):
    """r samples of mirror padding (the edge sample is not repeated: scipy's mode="mirror") on both ends of `axis`."""
    pads = [(0, 0)] * a.ndim
    pads[axis] = (r, r)
    return np.pad(a, pads, mode="reflect")


def gaussian_blur_u8(img_u8, sigma):
    """imgaug's blur_gaussian_ on a uint8 image = cv2.GaussianBlur(image, (k, k), sigma, borderType=BORDER_REFLECT_101) with the kernel
    size imgaug derives from sigma (3.3 sigma below 3, 2.9 sigma below 5, else 2.6 sigma; at least 5, made odd): a normalised
    exp(-x^2 / 2 sigma^2) kernel per axis, the 8-bit result rounded to nearest and saturated.  (cv2's 8-bit path quantises the kernel
    to 8 fractional bits; that last-level difference is not restated.  With the shipped parameter ranges sigma is always the floor
    value 0.5: k = 5.)"""
    sigma = float(sigma)
    k = 3.3 * sigma if sigma < 3.0 else (2.9 * sigma if sigma < 5.0 else 2.6 * sigma)
    k = int(max(k, 5))
    k += 1 - (k % 2)
    r = k // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    g /= g.sum()
    cur = np.asarray(img_u8, np.float64)
    for axis in (0, 1):
        p = mirror_pad(cur, r, axis)
        acc = np.zeros(cur.shape, np.float64)
        for j in range(k):
            sl = [slice(None)] * 2
            sl[axis] = slice(j, j + cur.shape[axis])
            acc += g[j] * p[tuple(sl)]
        cur = acc
    return np.minimum(np.maximum(np.floor(cur + 0.5), 0.0), 255.0).astype(np.uint8)


def correlate_mirror(img, kern):
    """scipy.ndimage.correlate(float image, kern, mode="mirror"): out[y, x] = sum kern[i, j] * img[y + i - kh // 2, x + j - kw // 2]."""
    img = np.asarray(img, np.float64)
    kh, kw = kern.shape
    oy, ox = kh // 2, kw // 2
    p = np.pad(img, ((oy, kh - 1 - oy), (ox, kw - 1 - ox)), mode="reflect")
    out = np.zeros(img.shape, np.float64)
    for i in range(kh):
        for j in range(kw):
            if kern[i, j] != 0.0:
                out += float(kern[i, j]) * p[i:i + img.shape[0], j:j + img.shape[1]]
    return out


def falling_noise(rs, h, w, spec, blur):
    """One layer's noise image (uint8 [h, w]), its speed and flake-size uniformity: salt on a canvas shrunk by the flake size, gated by a
    coarse Beta field, up-sampled, (snow) Gaussian-blurred, then smeared along the falling direction by a motion-blur kernel."""
    from oracle.datapipe_np import motion_blur_kernel
    flake = _param(rs, spec["flake_size"])
    uniformity = _param(rs, spec["flake_size_uniformity"])
    angle = _param(rs, spec["angle"])
    speed = _param(rs, spec["speed"])
    sigma_fraction = _param(rs, spec["blur_sigma_fraction"])
    density = _param(rs, spec["density"])
    density_uniformity = _param(rs, spec["density_uniformity"])
    shrink = min(max(1.0 - flake, 0.001), 1.0)
    hd, wd = max(1, int(h * shrink)), max(1, int(w * shrink))
    # Salt(p): a pixel of the black canvas is replaced, with probability p, by the upper half of 255 * Beta(0.5, 0.5)
    replaced = rs.rand(hd, wd) < density
    salt = np.abs(rs.beta(0.5, 0.5, size=(hd, wd)) - 0.5) + 0.5
    canvas = np.zeros((hd, wd), np.float64)
    canvas[replaced] = np.minimum(np.maximum(np.round(salt[replaced] * 255.0), 0.0), 255.0)
    gate = resize_cubic(rs.beta(1.0, max(1.0 - density_uniformity, 1e-6), size=(8, 8)), hd, wd)
    gated = np.floor(np.minimum(np.maximum(canvas * np.minimum(np.maximum(gate, 0.0), 1.0), 0.0), 255.0))   # (uint8 cast of a non-negative value)
    noise = resize_cubic(gated, h, w, to_u8=True).astype(np.uint8)
    if blur:
        sigma = min(max(max(h, w) * sigma_fraction, 0.5), 3.75)
        noise = gaussian_blur_u8(noise, sigma)
    k = int(speed * max(h, w))
    if k > 1:
        kern = motion_blur_kernel(max(k, 3), angle, 1.0).astype(np.float64)
        kern = kern / kern.sum()
        noise = np.minimum(np.maximum(np.round(correlate_mirror(noise, kern)), 0.0), 255.0).astype(np.uint8)
    return noise, speed, uniformity


def snowflake_layers(rs, h, w):
    """1 - 3 layers -> (what is added to the image, what the result is raised to at least)."""
    layers = []
    for _ in range(rs.randint(1, 4)):
        noise, speed, uniformity = falling_noise(rs, h, w, SNOWFLAKES, blur=True)
        gamma, regain = 1.0 + 2.0 * (1.0 - uniformity), 1.0 + 5.0 * (1.0 - uniformity)
        table = np.floor(255.0 * (np.arange(256) / 255.0) ** gamma)          # GammaContrast on uint8: a 256-entry table, truncated
        flakes = table[noise] * regain
        layers.append(((0.1 + 20.0 * speed) * flakes, (1.0 + 20.0 * speed) * flakes))
    return [(a.astype(np.float32), b.astype(np.float32)) for a, b in layers]


def rain_layers(rs, h, w):
    """1 - 3 layers -> (alpha = noise / 255, the drop colour: 110 + (240 - 110) % (sum of the first 1000 values of the RGB-tiled noise))."""
    layers = []
    for _ in range(rs.randint(1, 4)):
        noise, _, _ = falling_noise(rs, h, w, RAIN, blur=False)
        first = noise.reshape(-1)[:1000].astype(np.float64)
        total = 3.0 * float(first.sum())
        colour = 110.0 + float(np.fmod(130.0, total if total > 0 else 1.0))
        layers.append((noise.astype(np.float32) / 255.0, np.full((h, w), colour, np.float32)))
    return layers


# ------------------------------------------------------------------------------------------------ iaa.PiecewiseAffine
def piecewise_affine_map(rs, h, w, nb_rows=4, nb_cols=4, scale=(0.01, 0.1)):
    """-> (x source position [h, w], y source position [h, w]) of every output pixel.  A regular nb_rows x nb_cols mesh over the image
    is moved point by point by Normal(0, s) * (h, w) (one s per image, columns of the jitter are (y, x)) and clipped into the image;
    the transform takes every triangle of the REGULAR mesh onto its moved copy.  Inside a triangle an affine map IS barycentric
    interpolation of the moved corners, which is how it is evaluated here."""
    from scipy.spatial import Delaunay
    s = rs.uniform(scale[0], scale[1])
    jitter = rs.normal(0.0, s, size=(nb_rows * nb_cols, 2))
    gx, gy = np.linspace(0, w, nb_cols), np.linspace(0, h, nb_rows)
    mesh = np.array([(x, y) for y in gy for x in gx], np.float64)
    moved = np.empty_like(mesh)
    moved[:, 0] = np.minimum(np.maximum(mesh[:, 0] + jitter[:, 1] * w, 0.0), w - 1.0)
    moved[:, 1] = np.minimum(np.maximum(mesh[:, 1] + jitter[:, 0] * h, 0.0), h - 1.0)
    tri = Delaunay(mesh)
    sx, sy = np.full((h, w), -1.0), np.full((h, w), -1.0)
    for y in range(h):
        pts = np.stack([np.arange(w, dtype=np.float64), np.full(w, float(y))], 1)
        which = tri.find_simplex(pts)
        for x in range(w):
            k = which[x]
            if k < 0:
                continue
            a, b, c = (mesh[v] for v in tri.simplices[k])
            det = (b[1] - c[1]) * (a[0] - c[0]) + (c[0] - b[0]) * (a[1] - c[1])
            l0 = ((b[1] - c[1]) * (x - c[0]) + (c[0] - b[0]) * (y - c[1])) / det
            l1 = ((c[1] - a[1]) * (x - c[0]) + (a[0] - c[0]) * (y - c[1])) / det
            l2 = 1.0 - l0 - l1
            ma, mb, mc = (moved[v] for v in tri.simplices[k])
            sx[y, x] = l0 * ma[0] + l1 * mb[0] + l2 * mc[0]
            sy[y, x] = l0 * ma[1] + l1 * mb[1] + l2 * mc[1]
    return sx.astype(np.float32), sy.astype(np.float32)
