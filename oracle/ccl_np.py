"""ORACLE (test infrastructure only - never imported by the product path).

numpy restatement of the reference's host-side character labelling
    label_cluster.forward            /root/reference/Dino/utils/DBSCAN.py:65-103
which uses skimage.measure.label (8-connectivity, labels numbered in raster order of each component's
first pixel).  Pinned against tests/golden/ccl_cases.npz, produced by the real reference
(tools/gen_golden.py).

Algorithm restated:
  1. 8-connected components of mask != 0 (DBSCAN.py:79); label order = raster order of first pixel.
  2. walk the labels in that order, keep those with area >= 30, stop after 26 kept (DBSCAN.py:84-93).
  3. order the kept components by the mean column of their pixels (DBSCAN.py:89,94); the reference's
     np.argsort is unstable on ties - we break ties by label order (stable).
  4. emit one binary plane per kept component, in that order (DBSCAN.py:95-96).
The compact form used everywhere in this repo is an id map: uint8 [H, W], value = plane index, 255 = none.
"""
import numpy as np

BG = 255
MAX_COMPONENTS = 26
MIN_AREA = 30


def components_8conn(mask: np.ndarray) -> np.ndarray:
    """int32 [H,W]: 0 = background, else (raster index of the component's first pixel) + 1."""
    fg = np.asarray(mask) != 0
    h, w = fg.shape
    big = h * w + 1
    lab = np.where(fg, np.arange(1, h * w + 1, dtype=np.int32).reshape(h, w), big)
    while True:
        p = np.pad(lab, 1, constant_values=big)
        m = lab
        for dy in range(3):
            for dx in range(3):
                m = np.minimum(m, p[dy:dy + h, dx:dx + w])
        m = np.where(fg, m, big)
        if np.array_equal(m, lab):
            break
        lab = m
    return np.where(fg, lab, 0).astype(np.int32)


def label_idmap(mask: np.ndarray) -> np.ndarray:
    """mask [H,W] (nonzero = text) -> uint8 id map (plane index per pixel, 255 = background)."""
    comp = components_8conn(mask)
    out = np.full(comp.shape, BG, dtype=np.uint8)
    kept = []  # (label, column_sum, area)
    for lab in np.unique(comp):  # ascending = raster order of first pixel
        if lab == 0:
            continue
        ys, xs = np.nonzero(comp == lab)
        if xs.size >= MIN_AREA:
            kept.append((int(lab), int(xs.sum()), int(xs.size)))
            if len(kept) >= MAX_COMPONENTS:
                break
    # stable sort by mean column; compare exactly as rationals (equal to the float64 means' order)
    order = sorted(range(len(kept)), key=lambda i: (kept[i][1] / kept[i][2], i))
    for plane, i in enumerate(order):
        out[comp == kept[i][0]] = plane
    return out


def idmap_to_planes(idmap: np.ndarray, dtype=np.float32) -> np.ndarray:
    """[..., H, W] id map -> [..., 26, H, W] 0/1 planes (the reference's dense form)."""
    idmap = np.asarray(idmap)
    planes = (idmap[..., None, :, :] == np.arange(MAX_COMPONENTS, dtype=np.uint8)[:, None, None])
    return planes.astype(dtype)


def planes_to_idmap(planes: np.ndarray) -> np.ndarray:
    planes = np.asarray(planes) > 0
    cnt = planes.sum(axis=-3)
    if cnt.max() > 1:
        raise ValueError("planes overlap; id-map form would be lossy")
    ids = np.argmax(planes, axis=-3).astype(np.uint8)
    ids[cnt == 0] = BG
    return ids
