"""The pretraining dataset: word images + text masks from LMDB, three views per sample.

Drop-in for `Dino.dataset.datasetsupervised_kmeans.ImageDatasetSelfSupervisedKmeans` (reference :22-87 on top of
`ImageDataset`, Dino/dataset/dataset.py:17-167) with the work split the MI355X way: the host side (this class, inside
DataLoader workers) only reads the two LMDB records of a sample (`image-%09d` from `path`, `mask-%09d` from
`mask_path` + the part of `path` behind 'training', dataset.py:133-150), decodes them with PIL and resizes them to the
network resolution - 12 KiB of uint8 per sample; colour augmentation, the affine warp of view 2, normalisation and the
`theta` matrices are made per BATCH on the device by `DeviceViewMaker` (one kernel, csrc/kernels/datapipe.h).

    ds = ImageDatasetSelfSupervisedKmeans(path=..., mask_path=..., img_h=32, img_w=128, data_aug=True, augmentation_severity=5)
    loader = DataLoader(ds, batch_size=B, collate_fn=collate_uint8, ...)
    views = DeviceViewMaker(img_h=32, img_w=128, severity=5, seed=rank)
    for images_u8, masks in loader:
        image_tensors, masks, metrics = views(images_u8, masks)       # fp32 [B,3,3,32,128], [B,32,128], [B,3,3] on the GPU

The batch contract is the reference's (:82-87): views 0 = plain, 1 = colour, 2 = colour + warp (probability 0.7);
mask in {0,1} aligned with views 0/1 (cv2.resize + >= 0.5, :78-79); theta maps view-2 output coordinates to source
coordinates in the (size-1)-normalised convention of :65-71.
"""
from __future__ import annotations

import io
import os
import random
import warnings

import numpy as np
import torch
from torch.utils.data import Dataset

from . import lmdb_file
from .augment import P_W, sample_colour_params, sample_theta

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # dataset.py:79-80


def resize_bilinear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h)) with INTER_LINEAR: half-pixel centres, edge replication, no antialiasing.
    (float arithmetic + round-half-even; OpenCV's 11-bit fixed-point kernel can differ by one grey level.)"""
    img = np.asarray(img)
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    ys = (np.arange(out_h, dtype=np.float64) + 0.5) * (h / out_h) - 0.5
    xs = (np.arange(out_w, dtype=np.float64) + 0.5) * (w / out_w) - 0.5
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    ay = (ys - y0)[:, None]; ax = (xs - x0)[None, :]
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    src = img.astype(np.float64)
    if src.ndim == 3:
        ay, ax = ay[..., None], ax[..., None]
    top = src[y0c][:, x0c] * (1 - ax) + src[y0c][:, x1c] * ax
    bot = src[y1c][:, x0c] * (1 - ax) + src[y1c][:, x1c] * ax
    out = top * (1 - ay) + bot * ay
    if img.dtype == np.uint8:
        return np.rint(out).clip(0, 255).astype(np.uint8)
    return out.astype(img.dtype)


class ImageDatasetSelfSupervisedKmeans(Dataset):
    """Keyword surface of the reference class (the arguments train.py:400-433 passes); text-recognition arguments
    (charset, max_length, ...) are accepted and ignored - this dataset yields no labels (`mask=True` path, dataset.py:176-182)."""

    def __init__(self, path, is_training=True, img_h=32, img_w=128, convert_mode="RGB", data_aug=True, multiscales=False,
                 data_portion=1.0, mask=True, mask_path="", augmentation_severity=1, supervised_flag=False, **_ignored):
        self.path = os.fspath(path)
        if not os.path.isdir(self.path):
            raise AssertionError(f"{path} is not a valid directory.")
        if multiscales:
            raise NotImplementedError("multiscales=True (random aspect padding, dataset.py:93-123) is not used by the CCD configs")
        self.is_training, self.img_h, self.img_w = bool(is_training), int(img_h), int(img_w)
        self.convert_mode, self.data_aug, self.augmentation_severity = convert_mode, bool(data_aug), int(augmentation_severity)
        self._env = self._mask_env = None
        self.mask_dir = None
        if mask_path:
            # dataset.py:58-61: the mask environment mirrors the image tree below '.../training'
            parts = self.path.split("training")
            self.mask_dir = os.fspath(mask_path) + parts[1] if len(parts) > 1 else os.fspath(mask_path)
            if not os.path.isdir(self.mask_dir):
                raise AssertionError(f"mask LMDB {self.mask_dir} (for {self.path}) does not exist")
        with lmdb_file.LmdbReader(self.path) as env:
            n = env.get(b"num-samples")
            if n is None:
                raise lmdb_file.LmdbError(f"{self.path}: no 'num-samples' record")
            dataset_length = int(n)
        self.use_portion = self.is_training and data_portion != 1.0
        self.length = dataset_length if not self.use_portion else int(data_portion * dataset_length)
        if self.use_portion:
            self.optional_ind = np.random.permutation(dataset_length)[:self.length]

    def __len__(self):
        return self.length

    # LMDB readers are opened lazily, once per DataLoader worker process (an mmap does not survive pickling)
    def _envs(self):
        if self._env is None:
            self._env = lmdb_file.LmdbReader(self.path)
            self._mask_env = lmdb_file.LmdbReader(self.mask_dir) if self.mask_dir else None
        return self._env, self._mask_env

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_env"] = state["_mask_env"] = None
        return state

    def _decode(self, buf, mode):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)          # EXIF warning from TiffPlugin (dataset.py:139)
            from PIL import Image
            return Image.open(io.BytesIO(buf)).convert(mode)

    def get(self, idx, _depth=0):
        env, mask_env = self._envs()
        try:
            buf = env.get(f"image-{idx + 1:09d}".encode())
            image = np.asarray(self._decode(buf, self.convert_mode))
            if self.is_training and (image.shape[1] <= 6 or image.shape[0] <= 6):          # _check_image, dataset.py:88-92
                raise ValueError("image too small")
        except Exception:
            if not self.is_training or _depth > 16:
                return None
            nxt = random.randint(0, len(self) - 1)                 # _next_image, dataset.py:81-86
            return self.get(int(self.optional_ind[nxt]) if self.use_portion else nxt, _depth + 1)
        mask = None
        if mask_env is not None:
            try:
                mask = np.asarray(self._decode(mask_env.get(f"mask-{idx + 1:09d}".encode()), "L"))
            except Exception:
                mask = None
        if mask is None:
            mask = np.zeros(image.shape[:2], np.uint8)             # "Corrupted image" fallback, dataset.py:151-153
        return image, mask, idx

    def __getitem__(self, idx):
        if self.use_portion:
            idx = int(self.optional_ind[idx])
        datum = self.get(idx)
        if datum is None:
            return None
        image, mask, idx_new = datum
        image = resize_bilinear(image, self.img_h, self.img_w)
        mask_view = resize_bilinear(mask.astype(np.float32), self.img_h, self.img_w)
        mask_view = (mask_view >= 0.5).astype(np.float32)          # datasetsupervised_kmeans.py:78-79
        return torch.from_numpy(image), torch.from_numpy(mask_view)


def collate_uint8(batch):
    """collate_fn_filter_none (dataset.py:215-217) for (uint8 [H,W,3], float [H,W]) samples."""
    batch = [b for b in batch if b is not None]
    return torch.stack([b[0] for b in batch]), torch.stack([b[1] for b in batch])


class DeviceViewMaker:
    """(images uint8 [B,H,W,3], masks [B,H,W]) -> (image_tensors fp32 [B,3,3,H,W], masks fp32 [B,H,W], metrics fp32 [B,3,3])
    on the GPU: the per-batch half of `_process_training` (datasetsupervised_kmeans.py:48-81)."""

    def __init__(self, img_h=32, img_w=128, severity=5, data_aug=True, seed=0, device=None, workers=None):
        self.h, self.w, self.severity, self.data_aug = int(img_h), int(img_w), int(severity), bool(data_aug)
        self.workers, self._farm, self._ahead = workers, None, None      # weather-layer worker processes (None: half the cores, at most 32)
        self.rs = np.random.RandomState(seed)
        self.device = device

    def _draw(self, B):
        from .weather import LayerFarm, Overlays
        if self._farm is None:
            self._farm = LayerFarm(self.workers)
        theta, warped = sample_theta(self.rs, B, self.h, self.w, return_warped=True)
        overlays = Overlays(self.h, self.w, self._farm)
        params = sample_colour_params(self.rs, B, self.severity, warped=warped, h=self.h, w=self.w, overlays=overlays,
                                      resolve=False)               # view 2 = the plain image where not warped
        return B, theta, params, overlays

    def __call__(self, images_u8, masks):
        from .. import ops
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        B = images_u8.shape[0]
        assert tuple(images_u8.shape[1:]) == (self.h, self.w, 3) and images_u8.dtype == torch.uint8
        if self.data_aug and self.severity > 0:
            # The parameters of a batch do not depend on its images: they are drawn ONE CALL AHEAD, and the weather layers they ask
            # for (frequency-noise maps etc., ~1 ms of numpy each, ~500 per 256 samples) are computed by the worker pool while the
            # GPU runs the iteration in between.  A batch of another size (the last one of an epoch) is drawn on the spot.
            if self._ahead is None or self._ahead[0] != B:
                self._ahead = self._draw(B)
            _, theta, params, overlays = self._ahead
            planes = overlays.resolve(params, P_W)
            self._ahead = None
        else:                                                       # data_aug off: three identical views, identity theta
            params = sample_colour_params(self.rs, B, 0)
            theta = np.tile(np.eye(3, dtype=np.float32), (B, 1, 1))
            planes = None
        img_d = images_u8.to(dev, non_blocking=True).contiguous()
        out = ops.augment_views(img_d, torch.from_numpy(params).to(dev), torch.from_numpy(theta).to(dev), MEAN, STD,
                                overlay=None if planes is None else torch.from_numpy(planes).to(dev))
        result = out, masks.to(dev, non_blocking=True).float(), torch.from_numpy(theta).to(dev)
        if self.data_aug and self.severity > 0:
            self._ahead = self._draw(B)                             # (after the launches: the host draws while the device works)
        return result
