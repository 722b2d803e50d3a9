"""Seeded synthetic pretraining batches with the reference dataset's output contract.

The reference dataset (Dino/dataset/datasetsupervised_kmeans.py:48-87) yields per sample
  image_tensors float32 [3, 3, 32, 128]   views: 0 plain, 1 colour-augmented, 2 colour + affine warp
  mask          float32 [32, 128] in {0,1} aligned with views 0/1
  theta         float32 [3, 3]             normalised-coordinate affine, view-2 output -> source coords
This module fabricates batches of that shape: gaussian "images", masks made of
rectangular characters, and thetas drawn from the reference's augmentation ranges
(datasetsupervised_kmeans.py:40-45: scale 0.6-1.1, translate +-2 %, rotate +-10 deg, shear x +-45 deg /
y +-10 deg; identity with probability 0.3, :60).  Pure numpy/torch-CPU, deterministic for a given seed.
"""
from __future__ import annotations

import math

import numpy as np
import torch

IMG_H, IMG_W = 32, 128
CHAR_W, CHAR_H, CHAR_TOP = 10, 16, 8
X_SLOTS = tuple(range(2, 101, 14))  # 8 slots, >= 4 px gaps between characters


def _affine_theta(rs: np.random.RandomState) -> np.ndarray:
    """One view-2 theta in normalised coordinates (output grid -> source grid)."""
    if rs.uniform() <= 0.3:
        return np.eye(3, dtype=np.float32)
    sx, sy = rs.uniform(0.6, 1.1, size=2)
    tx, ty = rs.uniform(-0.02, 0.02, size=2) * (IMG_W, IMG_H)
    rot = math.radians(rs.uniform(-10.0, 10.0))
    shx = math.radians(rs.uniform(-45.0, 45.0))
    shy = math.radians(rs.uniform(-10.0, 10.0))
    cx, cy = (IMG_W - 1) / 2.0, (IMG_H - 1) / 2.0
    to_origin = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]], dtype=np.float64)
    from_origin = np.array([[1, 0, cx + tx], [0, 1, cy + ty], [0, 0, 1]], dtype=np.float64)
    scale = np.diag([sx, sy, 1.0])
    shear = np.array([[1, math.tan(shx), 0], [math.tan(shy), 1, 0], [0, 0, 1]], dtype=np.float64)
    rotm = np.array([[math.cos(rot), -math.sin(rot), 0], [math.sin(rot), math.cos(rot), 0], [0, 0, 1]])
    forward_px = from_origin @ rotm @ shear @ scale @ to_origin  # source px -> warped px
    inverse_px = np.linalg.inv(forward_px)  # warped px -> source px (what grid_sample wants)
    norm = np.array([[2.0 / (IMG_W - 1), 0, -1], [0, 2.0 / (IMG_H - 1), -1], [0, 0, 1]], dtype=np.float64)
    return (norm @ inverse_px @ np.linalg.inv(norm)).astype(np.float32)


def make_masks(batch: int, rs: np.random.RandomState) -> np.ndarray:
    masks = np.zeros((batch, IMG_H, IMG_W), dtype=np.float32)
    for b in range(batch):
        n_chars = rs.randint(3, 9)
        starts = rs.choice(len(X_SLOTS), size=n_chars, replace=False)
        for s in starts:
            x0 = X_SLOTS[s]
            masks[b, CHAR_TOP:CHAR_TOP + CHAR_H, x0:x0 + CHAR_W] = 1.0
    return masks


def make_batch(batch: int, seed: int = 0, device: str | torch.device = "cpu"):
    """Returns (image_tensors [B,3,3,32,128], masks [B,32,128], metrics [B,3,3]) as float32 tensors."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    images = torch.randn(batch, 3, 3, IMG_H, IMG_W, generator=gen, dtype=torch.float32)
    rs = np.random.RandomState(seed)
    masks = torch.from_numpy(make_masks(batch, rs))
    metrics = torch.from_numpy(np.stack([_affine_theta(rs) for _ in range(batch)]))
    return images.to(device), masks.to(device), metrics.to(device)


def make_text_like_batch(batch: int, seed: int = 0, amp: float = 2.0, noise: float = 0.3,
                         device: str | torch.device = "cpu"):
    """As make_batch, but the images CARRY the characters (bright rectangles on a dark ground + a little noise; view 2 is
    the affine warp of views 0/1 by the same theta the batch hands to the model).  On pure-noise images a randomly
    initialised segmentation head predicts salt and pepper - no component of 30 pixels survives - so the predicted-mask
    branch (epoch >= 30, dino_vision.py:64-70) is only exercised in earnest by images with structure."""
    import torch.nn.functional as F
    images, masks, metrics = make_batch(batch, seed)
    pattern = (masks * 2.0 - 1.0) * amp
    grid = F.affine_grid(metrics[:, :2, :], size=(batch, 1, IMG_H, IMG_W), align_corners=False)
    warped = F.grid_sample(pattern.unsqueeze(1), grid, align_corners=False).squeeze(1)
    images = images * noise
    images[:, 0] += pattern[:, None]
    images[:, 1] += pattern[:, None]
    images[:, 2] += warped[:, None]
    return images.to(device), masks.to(device), metrics.to(device)
