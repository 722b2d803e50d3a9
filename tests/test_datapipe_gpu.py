"""The data pipeline end to end on the GPU: image LMDB -> mask LMDB (device 2-means) -> dataset -> device view maker ->
the model's batch contract; then `train.py` on the reference-shaped YAML with `scheme: selfsupervised_kmeans` UNCHANGED."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from backends import Backend

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    with Backend("hip") as b:
        yield b


def word_png(rs, h, w):
    from PIL import Image
    dark_text = rs.uniform() < 0.5
    fg, bg = (rs.randint(10, 80), rs.randint(160, 245)) if dark_text else (rs.randint(170, 250), rs.randint(5, 90))
    img = np.full((h, w, 3), bg, np.float64)
    n = rs.randint(2, 8)
    cw = (w - 6) // n
    for c in range(n):
        x0 = 3 + c * cw
        img[h // 5: h - h // 5, x0: x0 + max(2, int(cw * 0.6))] = fg
    img += rs.normal(0, 6, size=img.shape)
    out = io.BytesIO()
    Image.fromarray(np.clip(np.rint(img), 0, 255).astype(np.uint8)).save(out, format="PNG")
    return out.getvalue()


def build_image_lmdb(root, n, seed=0):
    from ccd_amd.dataset import lmdb_file
    rs = np.random.RandomState(seed)
    recs = {b"num-samples": str(n).encode()}
    for i in range(1, n + 1):
        recs[b"image-%09d" % i] = word_png(rs, int(rs.randint(24, 64)), int(rs.randint(64, 256)))
        recs[b"label-%09d" % i] = b"text"
    lmdb_file.write_lmdb(root, recs)


def test_lmdb_to_batch_contract(hip, tmp_path):
    from PIL import Image
    from ccd_amd.dataset import DeviceViewMaker, ImageDatasetSelfSupervisedKmeans, collate_uint8, lmdb_file
    from ccd_amd.dataset.generate_mask import generate
    from oracle import datapipe_np as D
    root = str(tmp_path / "data_lmdb" / "training" / "label" / "Synth" / "MJ")
    mask_base = str(tmp_path / "data_lmdb" / "Mask")
    build_image_lmdb(root, 40)
    stat = generate(root, mask_base + "/label/Synth/MJ", batch=16, device=hip.device)
    assert stat["entries"] == 41
    with lmdb_file.LmdbReader(root) as env, lmdb_file.LmdbReader(mask_base + "/label/Synth/MJ") as menv:
        assert int(menv.get(b"num-samples")) == 40
        for i in (1, 17, 40):                                       # mask records == the oracle on PIL's gray image, as PNG 0/1
            gray = np.asarray(Image.open(io.BytesIO(env.get(b"image-%09d" % i))).convert("L"))
            m = np.asarray(Image.open(io.BytesIO(menv.get(b"mask-%09d" % i))))
            np.testing.assert_array_equal(m, D.kmeans2_mask(gray))
            assert 0.05 < m.mean() < 0.7                            # text is the minority cluster, whatever its polarity
    ds = ImageDatasetSelfSupervisedKmeans(path=root, mask_path=mask_base, img_h=32, img_w=128, augmentation_severity=5)
    loader = torch.utils.data.DataLoader(ds, batch_size=8, collate_fn=collate_uint8, num_workers=2, drop_last=True)
    views = DeviceViewMaker(32, 128, severity=5, seed=3)
    seen_warp = seen_id = False
    for images_u8, masks in loader:
        image_tensors, m, metrics = views(images_u8, masks)
        assert image_tensors.is_cuda and tuple(image_tensors.shape) == (8, 3, 3, 32, 128) and image_tensors.dtype == torch.float32
        assert tuple(m.shape) == (8, 32, 128) and tuple(metrics.shape) == (8, 3, 3)
        assert bool(torch.isfinite(image_tensors).all()) and float(image_tensors.abs().max()) < 3.0     # normalised images
        assert set(torch.unique(m).tolist()) <= {0.0, 1.0}
        ident = (metrics == torch.eye(3, device=metrics.device)).all(dim=(1, 2))
        seen_warp |= bool((~ident).any()); seen_id |= bool(ident.any())
        # view 0 is the plain normalised sample
        v0 = (images_u8.float().to(image_tensors.device) / 255.0 - torch.tensor([0.485, 0.456, 0.406], device=m.device)) \
            / torch.tensor([0.229, 0.224, 0.225], device=m.device)
        torch.testing.assert_close(image_tensors[:, 0], v0.permute(0, 3, 1, 2), rtol=0, atol=1e-5)
    assert seen_warp and seen_id


def test_finetune_dataset_augments_on_the_device(hip, tmp_path):
    """dataset.data_aug of the finetuning configs (CCD_vision_model_ARD.yaml:34; reference dataset_pretrain.py:68-158, 250-253):
    the labelled dataset hands out resized uint8 samples and DeviceImageAugmenter runs the pipeline per batch; without data_aug
    (and in evaluation) the host path returns the normalised tensor.  The two agree wherever the draw is the identity."""
    from ccd_amd.dataset import augment as A
    from ccd_amd.dataset.dataset_pretrain import DeviceImageAugmenter, ImageDataset, collate_fn_filter_none
    root = str(tmp_path / "labelled")
    build_image_lmdb(root, 24, seed=5)
    aug_ds = ImageDataset(path=root, is_training=True, data_aug=True)
    plain_ds = ImageDataset(path=root, is_training=True, data_aug=False)
    eval_ds = ImageDataset(path=root, is_training=False, data_aug=True)
    assert aug_ds.data_aug and not plain_ds.data_aug and not eval_ds.data_aug         # (:68 `is_training and data_aug`)
    u8, target = aug_ds[3]
    ref, _ = plain_ds[3]
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (32, 128, 3) and ref.dtype == torch.float32 and tuple(target.shape) == (1, 25)
    assert eval_ds[3][0].dtype == torch.float32
    loader = torch.utils.data.DataLoader(aug_ds, batch_size=8, collate_fn=collate_fn_filter_none, num_workers=2, drop_last=True)
    augment = DeviceImageAugmenter(32, 128, seed=11, device=hip.device)
    changed = 0
    for images_u8, labels in loader:
        out = augment(images_u8)
        assert out.is_cuda and tuple(out.shape) == (8, 3, 32, 128) and out.dtype == torch.float32 and bool(torch.isfinite(out).all())
        host = (images_u8.float() / 255.0 - torch.tensor(A_MEAN)) / torch.tensor(A_STD)
        changed += int(((out.cpu() - host.permute(0, 3, 1, 2)).abs().amax(dim=(1, 2, 3)) > 1e-3).sum())
    assert 8 <= changed <= 24                                         # most samples are augmented, some draws are the identity
    # an identity draw reproduces the host path exactly
    ident = np.tile(A.IDENTITY_PARAMS, (1, 2, 1)).astype(np.float32)
    from ccd_amd import ops
    got = ops.augment_views(u8[None].to(hip.device), torch.from_numpy(ident).to(hip.device),
                            torch.eye(3, device=hip.device)[None].contiguous(), A_MEAN, A_STD)[0, 2]
    torch.testing.assert_close(got.cpu(), ref, rtol=0, atol=1e-5)
    p, th = A.sample_finetune_params(np.random.RandomState(0), 2000, 32, 128)
    assert p.shape == (2000, 2, 96) and (p[:, 0] == A.IDENTITY_PARAMS).all() and np.isfinite(p).all()
    warped = ~(th == np.eye(3, dtype=np.float32)).all(axis=(1, 2))
    assert 0.3 < warped.mean() < 0.5                                  # 0.6 x (Affine | Rotate) of three geometric members
    assert (p[:, 1, A.P_C] == A.C_MEDIAN).any() and (p[:, 1, A.P_A] == A.A_JPEG).any() and (p[:, 1, A.P_C] != A.C_BILATERAL).all()   # no bilateral member here


A_MEAN, A_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def test_train_cli_on_lmdb_dataset(tmp_path):
    """The reference-shaped experiment file with its own `scheme: selfsupervised_kmeans`: only the paths are filled in."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29643", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    base = tmp_path / "data_lmdb"
    prep = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_datapipe_gpu import build_image_lmdb\n"
            "from ccd_amd.dataset.generate_mask import generate\n"
            "for k, sub in enumerate(['label/Synth/MJ', 'label/Synth/ST', 'URD/OCR-CC']):\n"
            "    build_image_lmdb(%r + '/training/' + sub, 96, seed=k)\n"
            "    generate(%r + '/training/' + sub, %r + '/Mask/' + sub)\n") % (REPO, os.path.join(REPO, "tests"), str(base), str(base), str(base))
    r = subprocess.run([sys.executable, "-c", prep], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    src = open(os.path.join(REPO, "Dino", "configs", "CCD_pretrain_ViT_Tiny.yaml")).read()
    assert "scheme: selfsupervised_kmeans" in src
    cfg = (src.replace("xxx/data_lmdb", str(base))
              .replace("imgnet_based: 1000000", "imgnet_based: 128")
              .replace("training: {epochs: 3,", "training: {epochs: 1,")
              .replace("num_workers: 16", "num_workers: 2")
              .replace("name: pre_tiny_65536", "name: lmdb_smoke"))
    import re
    cfg = re.sub(r"mask_path: '[^']*'", "mask_path: '%s/Mask/'" % str(base), cfg)
    cfg = re.sub(r"train: \{roots: \[[^\]]*\]", "train: {roots: ['%s/training/label/Synth/', '%s/training/URD/OCR-CC']" % (str(base), str(base)), cfg, flags=re.S)
    (tmp_path / "lmdb.yaml").write_text(cfg)
    run = subprocess.run([sys.executable, os.path.join(REPO, "train.py"), "--config", str(tmp_path / "lmdb.yaml")], cwd=tmp_path,
                         env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2500:] + run.stderr[-2500:]
    assert "Starting DINO training" in run.stdout and "Training time" in run.stdout
    ckpt = tmp_path / "saved_models" / "lmdb_smoke" / "checkpoint.pth"
    assert ckpt.is_file()
    sd = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert sd["iteration"] >= 2
    import json
    log = [json.loads(l) for l in open(tmp_path / "saved_models" / "lmdb_smoke" / "log.txt")]
    assert all(np.isfinite(e["train_loss"]) for e in log if "train_loss" in e)
