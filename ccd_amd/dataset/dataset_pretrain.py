"""Labelled word images from LMDB for the recogniser (finetuning / evaluation).

Drop-in for `Dino.dataset.dataset_pretrain.ImageDataset` (reference :18-277): records `image-%09d` / `label-%09d` /
`num-samples`; training samples are (normalised image fp32 [3,h,w], target indices [max_length] from AttnConvertor.str2tensor,
:215-222), evaluation samples are (image, [raw_label]) so that the default collate hands TextAccuracy a tuple of strings
in `label_tensors[0]` (eval_acc.py:37).  Empty labels and labels that encode to nothing are skipped in training like the
reference does (:213-221).  With `data_aug` (training only) a sample is the RESIZED UINT8 image and the finetuning-time imgaug
pipeline (:70-158) runs per batch on the device: `DeviceImageAugmenter` below (sampler: augment.sample_finetune_params; kernels:
csrc/kernels/datapipe.h) hands the model the normalised fp32 batch.  Without it: resize + ToTensor + ImageNet normalisation
on the host (:250-258)."""
from __future__ import annotations

import io
import os
import random
import warnings

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.dataloader import default_collate

from . import lmdb_file
from .datasetsupervised_kmeans import MEAN, STD, resize_bilinear


class ImageDataset(Dataset):
    def __init__(self, path, is_training=True, img_h=32, img_w=128, max_length=25, case_sensitive=False, type="DICT90",
                 convert_mode="RGB", data_aug=True, multiscales=False, data_portion=1.0, **_ignored):
        from ..convertor.attn import AttnConvertor
        self.path = os.fspath(path)
        if not os.path.isdir(self.path):
            raise AssertionError(f"{path} is not a valid directory.")
        if multiscales:
            raise NotImplementedError("multiscales=True is not used by the CCD configs")
        self.is_training, self.img_h, self.img_w, self.convert_mode = bool(is_training), int(img_h), int(img_w), convert_mode
        self.data_aug = bool(is_training) and bool(data_aug)                       # (:68 `if self.is_training and self.data_aug`)
        self.label_convertor = AttnConvertor(dict_type=type, max_seq_len=max_length, with_unknown=True)
        self._env = None
        with lmdb_file.LmdbReader(self.path) as env:
            dataset_length = int(env.get(b"num-samples"))
        print(f"current_dataset_path:{self.path}-->{dataset_length}")
        self.use_portion = self.is_training and data_portion != 1.0
        self.length = dataset_length if not self.use_portion else int(data_portion * dataset_length)
        if self.use_portion:
            self.optional_ind = np.random.permutation(dataset_length)[:self.length]
        self._mean = torch.tensor(MEAN).view(3, 1, 1)
        self._std = torch.tensor(STD).view(3, 1, 1)

    def __len__(self):
        return self.length

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_env"] = None
        return state

    def _next_image(self, depth):
        if not self.is_training or depth > 16:
            return None
        nxt = random.randint(0, len(self) - 1)
        return self.get(int(self.optional_ind[nxt]) if self.use_portion else nxt, depth + 1)

    def get(self, idx, depth=0):
        if self._env is None:
            self._env = lmdb_file.LmdbReader(self.path)
        try:
            raw_label = str(self._env.get(f"label-{idx + 1:09d}".encode()), "utf-8")
            if self.is_training and len(raw_label) == 0:
                return self._next_image(depth)
            if self.is_training:
                target = self.label_convertor.str2tensor([raw_label])
                if int(target[0][0]) == int(target[0][1]) == 91:           # '<BOS><EOS>': nothing encodable (:218-221)
                    return self._next_image(depth)
            else:
                target = [raw_label]
            from PIL import Image
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                image = Image.open(io.BytesIO(self._env.get(f"image-{idx + 1:09d}".encode()))).convert(self.convert_mode)
            if self.is_training and (image.size[0] <= 6 or image.size[1] <= 6):
                return self._next_image(depth)
        except Exception:
            return self._next_image(depth)
        return image, target

    def __getitem__(self, idx):
        if self.use_portion:
            idx = int(self.optional_ind[idx])
        datum = self.get(idx)
        if datum is None:
            return None
        image, text = datum
        arr = resize_bilinear(np.asarray(image), self.img_h, self.img_w)
        if self.data_aug:                     # uint8 [h, w, 3]: augmented + normalised per batch by DeviceImageAugmenter
            return torch.from_numpy(np.ascontiguousarray(arr)), text
        ten = torch.from_numpy(arr).permute(2, 0, 1).float().div(255.0)             # ToTensor
        return (ten - self._mean) / self._std, text


def collate_fn_filter_none(batch):
    batch = [b for b in batch if b is not None]
    return default_collate(batch)


class DeviceImageAugmenter:
    """images uint8 [B,h,w,3] -> normalised fp32 [B,3,h,w] on the GPU: `_process_training` of the labelled dataset
    (dataset_pretrain.py:250-253: augment_tfs, resize, ToTensor, normalize) with the augmentation after the resize, one
    launch pair per batch (ops.augment_views: the colour + warp view)."""

    def __init__(self, img_h=32, img_w=128, seed=0, device=None, workers=None):
        self.h, self.w, self.device = int(img_h), int(img_w), device
        self.workers, self._farm, self._ahead = workers, None, None
        self.rs = np.random.RandomState(seed)

    def _draw(self, B):
        from .augment import sample_finetune_params
        from .weather import LayerFarm, Overlays, WarpMaps
        if self._farm is None:
            self._farm = LayerFarm(self.workers)
        overlays, warps = Overlays(self.h, self.w, self._farm), WarpMaps(self.h, self.w, self._farm)
        params, theta = sample_finetune_params(self.rs, B, self.h, self.w, overlays=overlays, resolve=False, warps=warps)
        return B, params, theta, overlays, warps

    def __call__(self, images_u8):
        from .. import ops
        from .augment import P_W, P_WARP
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        B = images_u8.shape[0]
        assert tuple(images_u8.shape[1:]) == (self.h, self.w, 3) and images_u8.dtype == torch.uint8
        if self._ahead is None or self._ahead[0] != B:               # drawn one call ahead: the weather layers are computed by the
            self._ahead = self._draw(B)                              # worker pool while the GPU runs the iteration in between
        _, params, theta, overlays, warps = self._ahead
        planes, maps = overlays.resolve(params, P_W), warps.resolve(params, P_WARP)
        out = ops.augment_views(images_u8.to(dev, non_blocking=True).contiguous(), torch.from_numpy(params).to(dev),
                                torch.from_numpy(theta).to(dev), MEAN, STD,
                                overlay=None if planes is None else torch.from_numpy(planes).to(dev),
                                warp_maps=None if maps is None else torch.from_numpy(maps).to(dev))
        view = out[:, 2].contiguous()
        self._ahead = self._draw(B)                                  # (after the launches: the host draws while the device works)
        return view
