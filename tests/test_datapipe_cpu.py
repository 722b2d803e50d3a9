"""Host side of the data pipeline: the LMDB file format, the dataset's record handling and resize,
theta's composition, and the k-means oracle against the fixtures the REAL reference function wrote."""
import io
import os
import struct

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ccd_amd.dataset import lmdb_file
from ccd_amd.dataset.augment import affine_pixel_matrix, sample_colour_params, sample_theta, theta_from_pixel_matrix
from ccd_amd.dataset.datasetsupervised_kmeans import ImageDatasetSelfSupervisedKmeans, collate_uint8, resize_bilinear
from oracle import datapipe_np as D

HERE = os.path.dirname(os.path.abspath(__file__))


def _png(arr, mode=None):
    from PIL import Image
    out = io.BytesIO()
    Image.fromarray(arr, mode=mode).save(out, format="PNG")
    return out.getvalue()


@pytest.mark.parametrize("psize,count", [(4096, 3000), (512, 900)])
def test_lmdb_roundtrip(tmp_path, psize, count):
    rs = np.random.RandomState(1)
    items = {b"num-samples": str(count).encode()}
    for i in range(1, count + 1):
        items[b"image-%09d" % i] = rs.bytes(int(rs.choice([1, 17, 200, psize // 2 - 30, psize // 2, 3 * psize + 5])))
        items[b"label-%09d" % i] = b"w%d" % i
    stat = lmdb_file.write_lmdb(str(tmp_path / "db"), items, psize=psize)
    assert stat["entries"] == len(items) and stat["overflow_pages"] > 0
    assert stat["depth"] >= (3 if psize == 512 else 2)
    with lmdb_file.LmdbReader(str(tmp_path / "db")) as env:
        assert env.stat()["entries"] == len(items) and env.psize == psize and env.depth == stat["depth"]
        for k, v in items.items():
            assert env.get(k) == v, k
        for missing in (b"a", b"image-000000000", b"image-%09d" % (count + 1), b"zzzz", b"label", b"num-sample", b"num-samples0"):
            assert env.get(missing) is None
        assert [k for k, _ in env.items()] == sorted(items)
        with env.begin(write=False) as txn:                        # the reference's access pattern, dataset.py:65-66
            assert int(txn.get("num-samples".encode())) == count
    # the same through the lmdb.open(...) spelling
    env = lmdb_file.open(str(tmp_path / "db"), readonly=True, lock=False, readahead=False, meminit=False)
    assert env.get("label-000000007") == b"w7"
    env.close()


def test_lmdb_page_structure(tmp_path):
    """What liblmdb checks when it opens a file: magic / version / page size in meta 0, the newer meta wins, page numbers
    stored in the pages, branch page first key empty, leaf nodes sorted."""
    items = {b"k%05d" % i: b"v" * (i % 50) for i in range(2000)}
    lmdb_file.write_lmdb(str(tmp_path / "db"), items, psize=512)
    raw = open(tmp_path / "db" / "data.mdb", "rb").read()
    assert len(raw) % 512 == 0
    for pg in range(len(raw) // 512):
        pgno, _pad, flags = struct.unpack_from("<QHH", raw, pg * 512)
        if flags & lmdb_file.P_OVERFLOW and pgno != pg:
            continue                                               # inside an overflow run: data, not a header
        assert pgno == pg
    magic, version = struct.unpack_from("<II", raw, 16)
    assert magic == 0xBEEFC0DE and version == 1
    assert struct.unpack_from("<I", raw, 16 + 24)[0] == 512        # mm_dbs[FREE].md_pad = page size
    tx0 = struct.unpack_from("<Q", raw, 16 + 24 + 96 + 8)[0]
    tx1 = struct.unpack_from("<Q", raw, 512 + 16 + 24 + 96 + 8)[0]
    assert (tx0, tx1) == (0, 1)
    env = lmdb_file.LmdbReader(str(tmp_path / "db"))
    off, flags, lower, _ = env._page(env.root)
    assert flags & lmdb_file.P_BRANCH
    assert env._node(off, 0)[3] == 0                               # implicit lowest key
    env.close()


def test_lmdb_errors(tmp_path):
    with pytest.raises(lmdb_file.LmdbError):
        lmdb_file.LmdbReader(str(tmp_path / "nope"))
    os.makedirs(tmp_path / "bad")
    (tmp_path / "bad" / "data.mdb").write_bytes(b"\0" * 8192)
    with pytest.raises(lmdb_file.LmdbError, match="magic"):
        lmdb_file.LmdbReader(str(tmp_path / "bad"))
    with pytest.raises(ValueError):
        lmdb_file.write_lmdb(str(tmp_path / "dup"), [(b"a", b"1"), (b"a", b"2")])
    empty = lmdb_file.write_lmdb(str(tmp_path / "empty"), {})
    assert empty["entries"] == 0 and lmdb_file.LmdbReader(str(tmp_path / "empty")).get(b"x") is None


def test_kmeans_oracle_is_the_reference_function():
    """oracle/datapipe_np.kmeans2_mask == the outputs Dino/utils/kmeans.py::clusterpixels wrote (tools/gen_golden.py)."""
    z = np.load(os.path.join(HERE, "golden", "kmeans_masks.npz"))
    off = 0
    assert len(z["hw"]) >= 40
    for h, w in z["hw"]:
        g, m = z["gray"][off:off + h * w].reshape(h, w), z["mask"][off:off + h * w].reshape(h, w)
        off += h * w
        np.testing.assert_array_equal(D.kmeans2_mask(g), m)
    # PIL's "L" conversion (generate_mask.py:69)
    from PIL import Image
    rgb = np.random.RandomState(0).randint(0, 256, size=(9, 13, 3)).astype(np.uint8)
    np.testing.assert_array_equal(D.gray_from_rgb(rgb), np.asarray(Image.fromarray(rgb).convert("L")))


def test_theta_is_the_datasets_composition():
    """datasetsupervised_kmeans.py:63-70 with an original image of another size: metric = W_inv . M_orig^-1 . W,
    theta = W_ . metric . W_^-1.  The device augmenter draws M at network resolution; both must give the same theta for
    the same warp."""
    rs = np.random.RandomState(3)
    img_h, img_w, oh, ow = 32, 128, 47, 211
    for _ in range(20):
        m_net = affine_pixel_matrix(rs, img_h, img_w)
        w_scale, h_scale = ow / img_w, oh / img_h
        Wm = np.array([[w_scale, 0, 0], [0, h_scale, 0], [0, 0, 1]])
        W_inv = np.array([[1 / w_scale, 0, 0], [0, 1 / h_scale, 0], [0, 0, 1]])
        m_orig = Wm @ m_net @ W_inv                                 # the same warp expressed at the original resolution
        metric = W_inv @ np.linalg.inv(m_orig) @ Wm
        W_ = np.array([[2 / (img_w - 1), 0, -1], [0, 2 / (img_h - 1), -1], [0, 0, 1]])
        want = np.array(W_ @ metric @ np.linalg.inv(W_), dtype=np.float32)
        np.testing.assert_allclose(theta_from_pixel_matrix(m_net, img_h, img_w), want, rtol=1e-5, atol=1e-6)
    th = sample_theta(np.random.RandomState(0), 2000, img_h, img_w)
    ident = (th == np.eye(3, dtype=np.float32)).all(axis=(1, 2)).mean()
    assert 0.26 < ident < 0.34                                      # `random.random() > 0.3` -> warp
    assert np.allclose(th[:, 2], [0, 0, 1])
    from ccd_amd.dataset import augment as A
    th, warped = sample_theta(np.random.RandomState(3), 4000, img_h, img_w, return_warped=True)
    p = sample_colour_params(np.random.RandomState(0), 4000, 5, warped=warped, h=img_h, w=img_w)
    assert p.shape == (4000, 2, 96) and np.isfinite(p).all()
    ident = sample_colour_params(np.random.RandomState(0), 4, 0)
    assert (ident[..., 1:] == A.IDENTITY_PARAMS[1:]).all()                                       # severity 0: only the seed differs
    # a sample whose warp draw failed gets the plain image as view 2 (datasetsupervised_kmeans.py:72-74)
    assert (p[~warped, 1] == A.IDENTITY_PARAMS).all() and (p[warped, 1] != A.IDENTITY_PARAMS).any()
    v1 = p[:, 0]
    live = 0.8                                                       # Sometimes(0.2, Identity, Sequential[...])
    a_op, b_op, c_op, d_op = (v1[:, i].astype(int) for i in (A.P_A, A.P_B, A.P_C, A.P_D))
    share = lambda sel: sel.mean() / live
    # `arithmetic`: OneOf over the reference's 21 members (augmentation_pipelines.py:122-144), each 1 / 21 of the live draws
    want_a = {A.A_ADD_ELEM: 1, A.A_GAUSS: 1, A.A_LAPLACE: 1, A.A_POISSON: 1, A.A_MUL: 1, A.A_MUL_ELEM: 1, A.A_DROPOUT: 1, A.A_COARSE: 1,
              A.A_DROP2D: 1, A.A_REPLACE: 4, A.A_INVERT: 0.15, A.A_SOLARIZE: 0.5, A.A_JPEG: 1, A.A_FILTER: 3, A.A_PILFILTER: 2}
    for op, mult in want_a.items():
        assert abs(share(a_op == op) - mult / 21.0) < 0.02, (op, share(a_op == op), mult / 21.0)
    # `color`: Sometimes(0.7, OneOf 9)
    want_b = {A.B_HUE_ADD: 2, A.B_BRIGHT: 1, A.B_MUL_HS: 1, A.B_ADD_HS: 1, A.B_GRAY: 1, A.B_UNIFORM_Q: 1, A.B_GAINS: 1, A.B_KMEANS: 1}
    for op, mult in want_b.items():
        assert abs(share(b_op == op) - 0.7 * mult / 9.0) < 0.02, (op, share(b_op == op))
    km = v1[b_op == A.B_KMEANS]
    assert km[:, A.P_B + 1].min() >= 2 and km[:, A.P_B + 1].max() <= 16            # n_colors (2, 16)
    # `Blur`: Sometimes(0.7, OneOf[Sharpen, OneOf[5 blurs]]): filters = Sharpen + Gaussian + Average + Motion
    assert abs(share(c_op == A.C_FILTER) - 0.7 * (0.5 + 0.5 * 3 / 5)) < 0.03 and abs(share(c_op == A.C_MEDIAN) - 0.07) < 0.02
    assert abs(share(c_op == A.C_BILATERAL) - 0.07) < 0.02
    assert set(np.unique(v1[c_op == A.C_MEDIAN, A.P_C + 1])) == {3.0, 5.0, 7.0}
    bil = v1[c_op == A.C_BILATERAL]
    assert bil[:, A.P_C + 1].min() >= 3 and bil[:, A.P_C + 1].max() <= 10 and bil[:, A.P_C + 2].min() >= 10 and bil[:, A.P_C + 3].max() <= 250
    kern = v1[c_op == A.C_FILTER][:, A.P_KERN:A.P_KERN + 49]
    ks = kern.sum(1)                                                # the blurs preserve the mean; Sharpen(lightness 0 - 0.5) sums to 1 - alpha (1 - lightness)
    assert ks.max() < 1.0 + 1e-4 and ks.min() >= 0.5 and (np.abs(ks - 1.0) < 1e-4).mean() > 0.3
    # `contrast`: Sometimes(0.7, OneOf 8)
    for op in (A.D_GAMMA, A.D_LINEAR, A.D_SIGMOID, A.D_LOG, A.D_HISTEQ_ALL, A.D_HISTEQ_LAB, A.D_CLAHE_LAB, A.D_CLAHE_ALL):
        assert abs(share(d_op == op) - 0.7 / 8.0) < 0.02, (op, share(d_op == op))
    jq = v1[a_op == A.A_JPEG, A.P_A + 1]
    assert jq.min() >= 2 and jq.max() <= 31                         # compression 70-99 -> PIL quality 31 .. 2
    cd = v1[a_op == A.A_COARSE]
    assert (cd[:, A.P_A + 3] == 5).all() and (cd[:, A.P_A + 4] == 19).all() and (cd[:, A.P_A + 1] == np.float32(0.02)).all()
    hs = v1[b_op == A.B_ADD_HS]
    assert np.abs(hs[:, A.P_B + 1]).max() <= 35 and np.abs(hs[:, A.P_B + 2]).max() <= 50 and (hs[:, A.P_B + 1] == np.round(hs[:, A.P_B + 1])).all()
    # the finetuning pipeline (dataset_pretrain.py:80-146): 35-member OneOf behind Sometimes(0.8), leading Invert 0.6 * 0.1
    pf, thf = A.sample_finetune_params(np.random.RandomState(1), 6000, img_h, img_w)
    f1 = pf[:, 1]
    assert abs(f1[:, A.P_PREINV].mean() - 0.06) < 0.015
    fa, fb, fc = (f1[:, i].astype(int) for i in (A.P_A, A.P_B, A.P_C))
    assert abs((fa == A.A_JPEG).mean() - 0.8 / 35) < 0.01 and abs((fb == A.B_HUE_ADD).mean() - 0.8 * 2 / 35) < 0.012
    assert abs((fb == A.B_SHUFFLE).mean() - 0.8 * 0.35 / 35) < 0.006 and (fc == A.C_BILATERAL).sum() == 0
    assert ((fa != 0) & (fb != 0)).sum() == 0                       # ONE member of the combined list
    assert abs((thf != np.eye(3, dtype=np.float32)).any(axis=(1, 2)).mean() - 0.6 * 2 / 3) < 0.03      # PiecewiseAffine's third: identity


def test_member_restatements_of_the_colour_chain():
    """oracle/datapipe_np.py's pointwise / 3 x 3 members: the PIL filter presets against PIL itself; the restated cv2 conversions
    and tables against independent definitions (colorsys for HSV within its 8-bit quantisation, the textbook cumulative
    histogram for equalizeHist); DirectedEdgeDetect's kernel against the sampler's; the noise members' distributions."""
    import colorsys
    from PIL import Image, ImageFilter
    from ccd_amd.dataset import augment as A
    rs = np.random.RandomState(11)
    img = rs.randint(0, 256, size=(24, 56, 3)).astype(np.uint8)
    img[:8, :16] = (200, 30, 30)
    # pillike.FilterEdgeEnhanceMore / FilterContour ARE PIL's filters (kernel, scale, offset, copied border)
    for filt, kern, off in ((ImageFilter.EDGE_ENHANCE_MORE, [-1, -1, -1, -1, 9, -1, -1, -1, -1], 0.0),
                            (ImageFilter.CONTOUR, [-1, -1, -1, -1, 8, -1, -1, -1, -1], 255.0)):
        want = np.array(Image.fromarray(img).filter(filt))
        np.testing.assert_array_equal(D.filter3(img, kern, pil=True, offset=off, scale=1.0), want)
    # cv2's 8-bit HSV: H in units of 2 degrees (0..179), S and V on 0..255
    hsv = D.rgb_to_hsv_cv(img)
    for (y, x) in [(0, 0), (3, 40), (10, 20), (23, 55), (12, 7)]:
        h, s_, v = colorsys.rgb_to_hsv(*(img[y, x] / 255.0))
        dh = abs(hsv[y, x, 0] - h * 180.0)
        assert min(dh, 180 - dh) <= 1.0 and abs(hsv[y, x, 1] - s_ * 255.0) <= 1.0 and hsv[y, x, 2] == img[y, x].max()
    gray = np.repeat(rs.randint(0, 256, size=(4, 4, 1)), 3, axis=2).astype(np.uint8)
    assert (D.rgb_to_hsv_cv(gray)[..., :2] == 0).all()              # no hue / saturation on the grey axis
    back = D.hsv_to_rgb_cv(hsv)
    assert np.abs(back.astype(int) - img.astype(int)).max() <= 6    # H is quantised to 2 degrees
    np.testing.assert_array_equal(D.hsv_to_rgb_cv(D.rgb_to_hsv_cv(gray)), gray)
    shifted = D.colour_member(_row(b=A.B_ADD_HS, b_args=(90, 0)), img)            # + 180 degrees twice = the identity (up to quantisation)
    twice = D.colour_member(_row(b=A.B_ADD_HS, b_args=(90, 0)), shifted)
    assert np.abs(twice.astype(int) - img.astype(int)).max() <= 12
    # equalizeHist: monotone table, first occupied bin -> 0, last -> 255, a constant channel stays
    ch = np.clip(rs.normal(120, 20, size=(32, 128)), 0, 255).astype(np.uint8)
    eq = D.equalize_hist_cv(ch)
    lo, hi = ch.min(), ch.max()
    assert eq[ch == lo].max() == 0 and eq[ch == hi].min() == 255 and (np.diff(eq.reshape(-1)[np.argsort(ch.reshape(-1), kind="stable")]) >= 0).all()
    cdf = np.cumsum(np.bincount(ch.reshape(-1), minlength=256))
    textbook = np.rint((cdf - cdf[lo]) * 255.0 / (ch.size - cdf[lo]))
    np.testing.assert_array_equal(eq, textbook[ch].astype(np.uint8))
    np.testing.assert_array_equal(D.equalize_hist_cv(np.full((5, 5), 9, np.uint8)), np.full((5, 5), 9, np.uint8))
    # the contrast tables at their fixed points (imgaug's docs: 127 + alpha (v - 127); sigmoid(cutoff) = 1 / 2; log2(2) = 1)
    ramp = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    assert D.contrast_member(_row(d=A.D_LINEAR, d_args=(0.5,)), ramp)[0, 127, 0] == 127
    assert D.contrast_member(_row(d=A.D_LINEAR, d_args=(0.5,)), ramp)[0, 255, 0] == 191
    assert D.contrast_member(_row(d=A.D_SIGMOID, d_args=(10.0, 0.4)), ramp)[0, 102, 0] == 127          # 255 / 2, truncated
    assert D.contrast_member(_row(d=A.D_LOG, d_args=(1.0,)), ramp)[0, 255, 0] == 255
    assert D.contrast_member(_row(d=A.D_GAMMA, d_args=(2.0,)), ramp)[0, 128, 0] == int(255 * (128 / 255) ** 2)
    q = D.colour_member(_row(b=A.B_UNIFORM_Q, b_args=(4,)), ramp)[0, :, 0]
    assert sorted(set(q.tolist())) == [32, 96, 160, 224]            # bin centres
    # DirectedEdgeDetect: direction 0 looks up, 0.25 to the right; the sampler's kernel == the restatement's
    np.testing.assert_allclose(A.directed_edge_kernel(0.7, 0.3), D.directed_edge_kernel(0.7, 0.3), atol=1e-6)
    k0, k1 = D.directed_edge_kernel(1.0, 0.0), D.directed_edge_kernel(1.0, 0.25)
    assert k0[0, 1] == k0.min() and k1[1, 2] == k1.min() and abs(k0.sum()) < 1e-6 and k0[1, 1] == 1.0
    # the noise members: distributions of the counter-based draws (what the device kernel uses too)
    flat = np.full((64, 128, 3), 128, np.uint8)
    g = D.arith_pointwise(_row(seed=5, a=A.A_GAUSS, a_args=(20.0, 1)), flat).astype(float) - 128
    assert abs(g.mean()) < 0.5 and abs(g.std() - 20.0) < 0.6
    shared = D.arith_pointwise(_row(seed=5, a=A.A_GAUSS, a_args=(20.0, 0)), flat)
    assert (shared[..., 0] == shared[..., 1]).all() and (shared[..., 1] == shared[..., 2]).all()       # per_channel=False
    lap = D.arith_pointwise(_row(seed=6, a=A.A_LAPLACE, a_args=(10.0, 1)), flat).astype(float) - 128
    assert abs(lap.mean()) < 0.5 and abs(np.abs(lap).mean() - 10.0) < 0.5          # E|x| = scale
    poi = D.arith_pointwise(_row(seed=7, a=A.A_POISSON, a_args=(30.0, 1)), np.zeros_like(flat)).astype(float)
    assert abs(poi.mean() - 30.0) < 0.3 and abs(poi.var() - 30.0) < 2.0
    add = D.arith_pointwise(_row(seed=8, a=A.A_ADD_ELEM, a_args=(40, 1)), flat).astype(int) - 128
    assert add.min() == -40 and add.max() == 40 and abs(add.mean()) < 0.6
    drop = D.arith_pointwise(_row(seed=9, a=A.A_DROPOUT, a_args=(0.1, 0)), flat)
    assert abs((drop[..., 0] == 0).mean() - 0.1) < 0.01 and ((drop == 0).all(-1) == (drop == 0).any(-1)).all()
    coarse = D.arith_pointwise(_row(seed=10, a=A.A_COARSE, a_args=(0.3, 0, 5, 19)), flat)[..., 0] == 0
    cells = coarse.reshape(64, 128)
    assert 0.1 < cells.mean() < 0.5 and (cells[:12] == cells[0]).all()            # constant inside a coarse cell (64 / 5 rows)
    sp = D.arith_pointwise(_row(seed=11, a=A.A_REPLACE, a_args=(0.1, 0, 0)), flat)[..., 0].astype(int)
    hit = sp != 128
    assert abs(hit.mean() - 0.1) < 0.012 and (sp[hit] < 30).mean() > 0.18 and (sp[hit] > 225).mean() > 0.18   # the arcsine law: 0.22 in each tail (uniform: 0.12)
    assert D.arith_pointwise(_row(seed=11, a=A.A_REPLACE, a_args=(0.1, 0, 1)), flat).min() >= 127
    assert D.arith_pointwise(_row(seed=11, a=A.A_REPLACE, a_args=(0.1, 0, 2)), flat).max() <= 128
    np.testing.assert_array_equal(D.arith_pointwise(_row(a=A.A_DROP2D, a_args=(5,)), flat)[0, 0], [128, 0, 128])


def _row(**kw):
    from kernel_checks import _aug_params
    return _aug_params(**kw)


def test_spatial_member_restatements_are_pinned():
    """oracle/datapipe_np.py's neighbourhood members against the libraries imgaug delegates to, where this image has them:
    jpeg_roundtrip vs PIL / libjpeg's real encode + decode (iaa.JpegCompression IS PIL's JPEG), median_blur and filter7 vs
    scipy.ndimage; the motion-blur kernel of the sampler vs the restatement; reflect-101 borders."""
    import io
    from PIL import Image
    from scipy import ndimage
    from ccd_amd.dataset import augment as A
    rs = np.random.RandomState(7)

    def textlike(h, w):
        img = np.zeros((h, w, 3)) + rs.uniform(30, 220, 3)
        for _ in range(6):
            y0, x0 = rs.randint(0, h - 8), rs.randint(0, w - 8)
            img[y0:y0 + rs.randint(4, h // 2), x0:x0 + rs.randint(3, 12)] = rs.uniform(0, 255, 3)
        return np.clip(img + rs.normal(0, 6, img.shape), 0, 255).astype(np.uint8)

    for (h, w) in ((32, 128), (16, 40), (37, 101)):                   # whole MCUs, and sizes libjpeg pads by edge replication
        for q in (2, 5, 11, 20, 31):                                  # the qualities JpegCompression(70-99) maps to
            img = textlike(h, w)
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="JPEG", quality=q)
            ref = np.array(Image.open(io.BytesIO(buf.getvalue()))).astype(int)
            d = np.abs(D.jpeg_roundtrip(img, q).astype(int) - ref)
            # floating-point DCT here, libjpeg's 13-bit integer DCT there: a level or two, and rarely a coefficient that falls on
            # the other side of a quantisation boundary (one 8 x 8 block)
            assert d.mean() < 0.3 and (d > 2).mean() < 0.05, (h, w, q, d.max(), d.mean())
    assert D.jpeg_quality_from_compression(70) == 31 and D.jpeg_quality_from_compression(99) == 2
    img = textlike(32, 128)
    for k in (3, 5, 7):
        want = np.stack([ndimage.median_filter(img[..., c], size=k, mode="nearest") for c in range(3)], -1)
        np.testing.assert_array_equal(D.median_blur(img, k), want)
    kern = rs.rand(7, 7).astype(np.float32)
    kern /= kern.sum()
    want = np.stack([ndimage.correlate(img[..., c].astype(np.float32), kern, mode="mirror") for c in range(3)], -1)
    assert np.abs(D.filter7(img, kern).astype(int) - np.clip(np.floor(want + 0.5), 0, 255)).max() <= 1
    np.testing.assert_array_equal(D.reflect101(np.arange(-3, 8), 5), [3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1])
    for ang, dirn in ((0, 0), (45, 0.5), (133, -0.7), (270, 1.0)):
        np.testing.assert_allclose(A.motion_kernel(5, ang, dirn), D.motion_blur_kernel(5, ang, dirn), atol=1e-6)
    np.testing.assert_allclose(A.motion_kernel(5, 0, 0.0)[:, 2], 0.2, atol=1e-6)          # direction 0: a uniform vertical line
    # bilateral: a flat image is a fixed point; a hard edge with a small colour sigma stays an edge, with a large one it blurs
    flat = np.full((8, 8, 3), 77, np.uint8)
    np.testing.assert_array_equal(D.bilateral_blur(flat, 5, 30, 30), flat)
    edge = np.zeros((8, 16, 3), np.uint8)
    edge[:, 8:] = 200
    np.testing.assert_array_equal(D.bilateral_blur(edge, 5, 10, 50), edge)
    assert 0 < D.bilateral_blur(edge, 5, 250, 50)[4, 7, 0] < 200


def test_reader_on_a_hand_assembled_environment():
    """tests/golden/lmdb_handmade/data.mdb was assembled page by page by tools/make_lmdb_fixture.py - an independent code path
    from write_lmdb: nodes in insertion (not key) order from the page end downwards, an overflow run, a branch root whose first
    key is empty, the NEWER transaction on meta page 1 with a stale older tree and a garbage page left in the file.  (Still not
    a file liblmdb itself wrote: py-lmdb is not part of this image.)"""
    import hashlib
    import json
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lmdb_handmade")
    exp = json.load(open(os.path.join(root, "expected.json")))
    with lmdb_file.LmdbReader(root) as env:
        assert (env.txnid, env.depth, env.entries, env.psize) == (exp["txnid"], exp["depth"], exp["entries"], exp["psize"])
        assert env.overflow_pages == 3 and env.branch_pages == 1 and env.leaf_pages == 2
        for k, v in exp["records"].items():
            got = env.get(k.encode())
            assert got is not None and len(got) == v["len"] and hashlib.sha256(got).hexdigest() == v["sha256"], k
        assert [k.decode() for k, _ in env.items()] == sorted(exp["records"])
        assert env.get(b"num-samples") == b"3"                          # not the stale transaction's b"1"
        assert env.get(b"label-000000002").decode("utf-8") == "na\u00efve" and env.get(b"label-000000003") == b""
        for missing in (b"a", b"image-000000004", b"label-0000000015", b"zzz"):
            assert env.get(missing) is None
        from PIL import Image
        assert Image.open(io.BytesIO(env.get(b"image-000000002"))).size == (200, 48)


def test_streaming_writer_equals_the_bulk_load(tmp_path):
    """write_lmdb(presorted=True) consumes an iterator, writes pages as they fill (overflow runs in front of the leaf that
    points at them) and produces the same environment as the in-memory load; keys out of order are refused."""
    rs = np.random.RandomState(1)
    recs = {b"mask-%09d" % i: bytes(rs.randint(0, 256, size=int(rs.choice([40, 300, 2100, 9000]))).astype(np.uint8)) for i in range(1, 400)}
    recs[b"num-samples"] = b"399"
    a = lmdb_file.write_lmdb(str(tmp_path / "bulk"), recs)
    b = lmdb_file.write_lmdb(str(tmp_path / "stream"), iter(sorted(recs.items())), presorted=True)
    assert a == b and a["depth"] >= 2 and a["overflow_pages"] > 0
    assert open(tmp_path / "bulk" / "data.mdb", "rb").read() == open(tmp_path / "stream" / "data.mdb", "rb").read()
    with lmdb_file.LmdbReader(str(tmp_path / "stream")) as env:
        assert env.entries == 400 and all(env.get(k) == v for k, v in recs.items())
    with pytest.raises(ValueError):
        lmdb_file.write_lmdb(str(tmp_path / "bad"), iter([(b"b", b"1"), (b"a", b"2")]), presorted=True)


def test_resize_is_cv2_inter_linear_geometry():
    """Half-pixel centres, no antialiasing: torch's bilinear (align_corners=False) is the same map."""
    rs = np.random.RandomState(0)
    for (h, w) in [(47, 211), (20, 64), (32, 128), (64, 300), (9, 31)]:
        img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        got = resize_bilinear(img, 32, 128)
        want = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(32, 128), mode="bilinear",
                             align_corners=False)[0].permute(1, 2, 0).numpy()
        assert got.dtype == np.uint8 and np.abs(got.astype(np.float64) - want).max() <= 0.5 + 1e-9


def _make_lmdbs(tmp_path, n=12):
    rs = np.random.RandomState(5)
    root = tmp_path / "data" / "training" / "label" / "Synth" / "MJ"
    mask_root = tmp_path / "Mask"
    imgs, recs, mrecs = [], {}, {}
    for i in range(1, n + 1):
        h, w = int(rs.randint(16, 60)), int(rs.randint(40, 220))
        img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        imgs.append(img)
        recs[b"image-%09d" % i] = _png(img)
        recs[b"label-%09d" % i] = b"word"
        mrecs[b"mask-%09d" % i] = _png(D.kmeans2_mask(D.gray_from_rgb(img)), mode="L")
    recs[b"image-%09d" % 3] = b"not a png"                          # corrupted record -> another sample is drawn
    del mrecs[b"mask-%09d" % 5]                                      # missing mask -> zeros
    recs[b"num-samples"] = str(n).encode(); mrecs[b"num-samples"] = str(n).encode()
    lmdb_file.write_lmdb(str(root), recs)
    lmdb_file.write_lmdb(str(mask_root) + "/label/Synth/MJ", mrecs)
    return str(root), str(mask_root), imgs


def test_dataset_records_and_collate(tmp_path):
    root, mask_root, imgs = _make_lmdbs(tmp_path)
    ds = ImageDatasetSelfSupervisedKmeans(path=root, mask_path=mask_root, img_h=32, img_w=128, is_training=True,
                                          data_aug=True, augmentation_severity=5, charset_path="ignored", max_length=25)
    assert len(ds) == 12
    image, mask = ds[0]
    assert image.dtype == torch.uint8 and tuple(image.shape) == (32, 128, 3) and tuple(mask.shape) == (32, 128)
    np.testing.assert_array_equal(image.numpy(), resize_bilinear(imgs[0], 32, 128))
    want_mask = (resize_bilinear(D.kmeans2_mask(D.gray_from_rgb(imgs[0])).astype(np.float32), 32, 128) >= 0.5)
    np.testing.assert_array_equal(mask.numpy(), want_mask.astype(np.float32))
    assert set(np.unique(mask.numpy())) <= {0.0, 1.0}
    assert ds[2] is not None and tuple(ds[2][0].shape) == (32, 128, 3)        # corrupted -> replacement sample
    assert float(ds[4][1].abs().sum()) == 0.0                                 # no mask record -> zero mask
    loader = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=collate_uint8, num_workers=2, shuffle=False)
    batches = list(loader)
    assert len(batches) == 3 and tuple(batches[0][0].shape) == (4, 32, 128, 3) and batches[0][1].dtype == torch.float32
    ds_eval = ImageDatasetSelfSupervisedKmeans(path=root, mask_path=mask_root, img_h=32, img_w=128, is_training=False)
    assert ds_eval[2] is None and collate_uint8([ds_eval[1], ds_eval[2]])[0].shape[0] == 1
    half = ImageDatasetSelfSupervisedKmeans(path=root, mask_path=mask_root, img_w=128, data_portion=0.5)
    assert len(half) == 6
    with pytest.raises(AssertionError):
        ImageDatasetSelfSupervisedKmeans(path=root + "_missing", mask_path=mask_root)


def test_cloud_layers_host_side():
    """ccd_amd/dataset/weather.py (imgaug's FrequencyNoise / CloudLayer / Fog / Clouds restated): the bicubic resize against torch's
    (the same A = -0.75 kernel, half-pixel centres, clamped border), the noise's normalisation and spectrum slope, the maps' ranges
    and the members' share of the draw in the sampler."""
    import torch
    import torch.nn.functional as F
    from ccd_amd.dataset import augment as A, weather as Wt
    rs = np.random.RandomState(5)
    src = rs.rand(8, 8)
    want = F.interpolate(torch.from_numpy(src)[None, None], size=(32, 128), mode="bicubic", align_corners=False)[0, 0].numpy()
    np.testing.assert_allclose(Wt.resize_cubic(src, 32, 128), want, atol=1e-6)
    small = rs.rand(4, 7)
    want = F.interpolate(torch.from_numpy(small)[None, None], size=(16, 40), mode="bicubic", align_corners=False)[0, 0].numpy()
    np.testing.assert_allclose(Wt.resize_cubic(small, 16, 40), want, atol=1e-6)
    # full-resolution frequency noise: exactly 0..1, and smoother (more low-frequency energy) for a more negative exponent
    n_a, n_b = Wt.frequency_noise(rs, 32, 128, -1.0, 128), Wt.frequency_noise(rs, 32, 128, -3.0, 128)
    for n in (n_a, n_b):
        assert n.shape == (32, 128) and n.dtype == np.float32 and n.min() == 0.0 and n.max() == 1.0
    rough = lambda n: np.abs(np.diff(n, axis=1)).mean()
    assert rough(n_b) < 0.5 * rough(n_a)
    low = Wt.frequency_noise(rs, 32, 128, -2.0, 5)                     # generated at 4 x 5, up-sampled through 8 bits
    assert low.shape == (32, 128) and 0.0 <= low.min() and low.max() <= 1.0 and rough(low) < 0.02
    for layers, lo_mean in ((Wt.fog_layers(rs, 32, 128), 220.0), (Wt.clouds_layers(rs, 32, 128), 150.0)):
        assert 1 <= len(layers) <= 2
        for m in layers:
            assert m.shape == (2, 32, 128) and 0.0 <= m[0].min() and m[0].max() <= 1.0 and 0.0 <= m[1].min() and m[1].max() <= 255.0
            assert m[1].mean() > lo_mean - 60.0
    fog_alpha = np.mean([Wt.fog_layers(rs, 32, 128)[0][0].mean() for _ in range(40)])
    assert 0.3 < fog_alpha < 0.8, fog_alpha                                # (0.7..0.9 + 0.3 n) ** 0.9 * (0.4..0.9)
    # the sampler: Sometimes(0.8) x Sometimes(0.7, OneOf 4 weather members) = 56 % of the rows carry layers; without a collector none does
    ov = Wt.Overlays(32, 128)
    p = A.sample_colour_params(np.random.RandomState(3), 600, 5, overlays=ov)
    share = (p[:, :, A.P_W] > 0).mean()
    assert 0.50 < share < 0.62, share
    snow = (p[:, :, A.P_W + 2] == Wt.SNOW_MODE) & (p[:, :, A.P_W] > 0)
    assert 0.10 < snow.mean() < 0.18                                     # Snowflakes: a quarter of them
    planes = ov.planes()
    assert planes.dtype == np.float16 and planes.shape[1:] == (2, 32, 128)
    rows = p.reshape(-1, A.AUG_NP)
    used = rows[rows[:, A.P_W] > 0]
    assert (used[:, A.P_W + 1] + used[:, A.P_W] <= len(planes)).all() and len(planes) == int(used[:, A.P_W].sum())
    assert (A.sample_colour_params(np.random.RandomState(3), 200, 5)[:, :, A.P_W] == 0).all()
    pf, _ = A.sample_finetune_params(np.random.RandomState(1), 800, 32, 128, overlays=Wt.Overlays(32, 128))
    assert 0.06 < (pf[:, 1, A.P_W] > 0).mean() < 0.12                  # 0.8 * 4 / 35


def test_snow_rain_layers_and_the_layer_farm():
    """Snowflakes / Rain (imgaug SnowflakesLayer / RainLayer restated in weather.py): layer counts, flake statistics, the blend modes; the
    worker pool returns exactly what an in-place draw returns (a layer is a function of (member, seed) only), and a batch drawn one call
    ahead (resolve=False ... resolve()) equals the batch drawn at once."""
    from ccd_amd.dataset import augment as A, weather as Wt
    rs = np.random.RandomState(11)
    for name, lo, hi in (("Snowflakes", 1, 3), ("Rain", 1, 3)):
        counts = set()
        for _ in range(12):
            layers = Wt.MAKERS[name][0](rs, 32, 128)
            counts.add(len(layers))
            assert lo <= len(layers) <= hi
            for m in layers:
                assert m.shape == (2, 32, 128) and np.isfinite(m).all()
                if name == "Rain":                                       # (alpha, drop colour 110..240)
                    assert 0.0 <= m[0].min() and m[0].max() <= 1.0 and 110.0 <= m[1].min() and m[1].max() <= 240.0
                    assert m[0].mean() < 0.6                             # drops are sparse: most of the image shows through
                else:                                                    # (what is added, the floor): floor = (1 + 20 s) / (0.1 + 20 s) x added
                    assert m[0].min() >= 0.0 and (m[1] >= m[0]).all() and m[1].max() < 6e4      # fits fp16
                    assert (m[0] == 0).mean() > 0.2                      # flakes are sparse
        assert len(counts) >= 2, (name, counts)
    tasks = [(n, 100 + i, 32, 128) for i, n in enumerate(("Fog", "Clouds", "Snowflakes", "Rain") * 3)]
    direct = [Wt.draw_layers(t) for t in tasks]
    farm = Wt.LayerFarm(2)
    try:
        pooled = farm.submit(tasks).get(timeout=120)
        for a, b in zip(direct, pooled):
            assert a.dtype == np.float16 and a.shape == b.shape and (a == b).all()
        ov_a, ov_b = Wt.Overlays(32, 128), Wt.Overlays(32, 128, farm)
        pa = A.sample_colour_params(np.random.RandomState(4), 24, 5, overlays=ov_a)
        pb = A.sample_colour_params(np.random.RandomState(4), 24, 5, overlays=ov_b, resolve=False)
        assert (pb[:, :, A.P_W] <= 0).all() and (pb[:, :, A.P_W] < 0).any()          # task ids parked in the rows
        planes_b = ov_b.resolve(pb, A.P_W)
        assert (pa == pb).all() and (ov_a.planes() == planes_b).all()
    finally:
        farm.close()


def test_lab_and_clahe_restatements():
    """oracle/datapipe_np.py: 8-bit Lab against the published values of the primaries (CIE L*a*b*, D65: red 53.24 / 80.09 / 67.20 ...), the
    round trip's quantisation, CLAHE's limits (a huge clip limit = per-tile equalisation; output uses the range; a flat image stays flat),
    and the sampler: every contrast member now carries parameters."""
    from oracle import datapipe_np as D
    from ccd_amd.dataset import augment as A
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0], [128, 128, 128]]], np.uint8)
    np.testing.assert_array_equal(D.rgb_to_lab_u8(px)[0], [[255, 128, 128], [136, 208, 195], [224, 42, 211], [82, 207, 20], [0, 128, 128], [137, 128, 128]])
    back = D.lab_to_rgb_u8(D.rgb_to_lab_u8(px)).astype(int)
    assert np.abs(back - px.astype(int)).max() <= 7                      # saturated primaries sit on the gamut's edge: 8-bit Lab is lossy there
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (32, 128, 3)).astype(np.uint8)
    rt = D.lab_to_rgb_u8(D.rgb_to_lab_u8(img)).astype(int)
    assert np.abs(rt - img.astype(int)).mean() < 1.0
    ch = (rs.rand(32, 128) * 60 + 90).astype(np.uint8)                   # a low-contrast channel
    out = D.clahe_cv(ch, 40.0, 4)
    assert out.max() - out.min() > 200 and np.unique(D.clahe_cv(np.full((32, 128), 77, np.uint8), 2.0, 4)).size == 1
    mild = D.clahe_cv(ch, 0.1, 4)                                        # clip limit 1 count per bin: nearly the identity ramp
    assert np.abs(mild.astype(int) - ch.astype(int)).mean() < np.abs(out.astype(int) - ch.astype(int)).mean()
    assert D.clahe_cv(ch, 3.0, 12).shape == ch.shape and D.clahe_cv(ch[:16, :40], 3.0, 7).shape == (16, 40)     # padded grids
    p = A.sample_colour_params(np.random.RandomState(2), 3000, 5).reshape(-1, A.AUG_NP)
    ops_d = p[:, A.P_D].astype(int)
    drawn = ops_d[ops_d > 0]
    share = np.bincount(drawn, minlength=9)[1:] / len(drawn)
    assert (np.abs(share - 1 / 8) < 0.03).all(), share                    # OneOf 8: every member one eighth of the contrast draws
    cl = p[np.isin(ops_d, (A.D_CLAHE_LAB, A.D_CLAHE_ALL))]
    assert (cl[:, A.P_D + 1] >= 0.1).all() and (cl[:, A.P_D + 1] <= 8.0).all() and set(np.unique(cl[:, A.P_D + 2]).astype(int)) <= set(range(3, 13))


def test_piecewise_affine_maps():
    """weather.piecewise_affine_map (imgaug PiecewiseAffine through skimage's PiecewiseAffineTransform, restated): the per-triangle affine
    maps against an independent formulation (barycentric interpolation of the moved vertices in scipy's triangulation), the mesh
    points themselves, zero jitter = the identity, continuity, the clip into the image, and the member's share of the finetuning draw."""
    from scipy.spatial import Delaunay
    from ccd_amd.dataset import augment as A, weather as Wt
    h, w = 32, 128
    got = Wt.piecewise_affine_map(np.random.RandomState(7), h, w)[0]
    assert got.shape == (2, h, w) and got.dtype == np.float32
    # the same draws, interpolated barycentrically
    rs = np.random.RandomState(7)
    s = rs.uniform(0.01, 0.1)
    jitter = rs.normal(0.0, s, size=(16, 2))
    xx, yy = np.meshgrid(np.linspace(0, w, 4), np.linspace(0, h, 4))
    src = np.stack([xx.ravel(), yy.ravel()], 1)
    dst = src + jitter[:, ::-1] * np.array([w, h], np.float64)
    dst[:, 0], dst[:, 1] = np.clip(dst[:, 0], 0, w - 1), np.clip(dst[:, 1], 0, h - 1)
    tri = Delaunay(src)
    py, px = np.mgrid[0:h, 0:w]
    pts = np.stack([px.ravel(), py.ravel()], 1).astype(np.float64)
    simp = tri.find_simplex(pts)
    assert (simp >= 0).all()
    tr = tri.transform[simp]                                              # barycentric coordinates of every pixel in its triangle
    lam = np.einsum("nij,nj->ni", tr[:, :2], pts - tr[:, 2])
    lam = np.hstack([lam, 1.0 - lam.sum(1, keepdims=True)])
    want = np.einsum("ni,nid->nd", lam, dst[tri.simplices[simp]]).T.reshape(2, h, w)
    np.testing.assert_allclose(got, want, atol=2e-4)
    assert got[0].min() >= -1e-4 and got[0].max() <= w - 1 + 1e-4 and got[1].min() >= -1e-4 and got[1].max() <= h - 1 + 1e-4
    assert abs(got[0, 0, 0] - dst[0, 0]) < 1e-4 and abs(got[1, 0, 0] - dst[0, 1]) < 1e-4      # pixel (0, 0) IS a mesh point
    assert np.abs(np.diff(got[0], axis=1)).max() < 6.0 and np.abs(np.diff(got[1], axis=0)).max() < 6.0   # continuous across the triangles
    ident = Wt.piecewise_affine_map(np.random.RandomState(1), h, w, scale=(0.0, 0.0))[0]
    # (zero jitter: the identity, except in the last mesh column / row, whose points at x = w / y = h are clipped to w - 1 / h - 1 -
    # the library skips the warp altogether when every jitter is zero)
    np.testing.assert_allclose(ident[0][:, :2 * w // 3], px[:, :2 * w // 3], atol=1e-4)
    np.testing.assert_allclose(ident[1][:2 * h // 3], py[:2 * h // 3], atol=1e-4)
    assert np.abs(ident[0] - px).max() <= 1.0 and np.abs(ident[1] - py).max() <= 1.0
    # the sampler: Sometimes(0.6) x OneOf 3 = 20 % of the samples carry a map; theta stays the identity for them
    wm = Wt.WarpMaps(h, w)
    p, th = A.sample_finetune_params(np.random.RandomState(2), 400, h, w, warps=wm)
    rows = p[:, 1, A.P_WARP]
    assert 0.14 < (rows > 0).mean() < 0.26 and (rows >= 0).all()
    maps = wm.planes()
    assert maps.shape == (int((rows > 0).sum()), 2, h, w) and sorted(rows[rows > 0].astype(int)) == list(range(1, len(maps) + 1))
    assert (th[rows > 0] == np.eye(3, dtype=np.float32)).all()
    p0, _ = A.sample_finetune_params(np.random.RandomState(2), 400, h, w)       # without a collector: unwarped, the same other draws
    assert (p0[:, 1, A.P_WARP] == 0).all() and (np.delete(p0, A.P_WARP, axis=2) == np.delete(p, A.P_WARP, axis=2)).all()


def test_host_drawn_layers_against_their_second_restatement():
    """What draws the weather planes and the PiecewiseAffine maps (ccd_amd/dataset/weather.py) against oracle/weather_np.py - a second
    restatement of the same published algorithms in other formulations (dense resampling matrices, fftfreq distances and one inverse
    FFT per axis, explicit mirror padding and window sums, barycentric interpolation) - value by value on shared seeds: the product's
    generators are not their own checker any more.  Float fields agree to rounding; layers that pass through 8-bit images (the cubic
    up-sampling, the blurs) may differ by ONE level where the two formulations round a .5 apart - on at most 0.2 % of a layer's pixels.
    Building blocks that a library of this image does implement are pinned to it (scipy.ndimage.correlate, torch's bicubic kernel)."""
    from scipy import ndimage
    from ccd_amd.dataset import weather as Wt
    from oracle import weather_np as On
    h, w = 32, 128
    rs = np.random.RandomState(0)
    img = (rs.rand(h, w) * 255).astype(np.uint8)
    for shape in ((5, 5), (4, 6), (3, 7)):
        k = rs.rand(*shape)
        assert np.abs(On.correlate_mirror(img, k) - ndimage.correlate(img.astype(np.float64), k, mode="mirror")).max() < 1e-9
    src = rs.rand(7, 9)
    assert np.abs(On.resize_cubic(src, h, w) - Wt.resize_cubic(src, h, w)).max() < 1e-12
    ref = F.interpolate(torch.from_numpy(src)[None, None], size=(h, w), mode="bicubic", align_corners=False)[0, 0].numpy()
    assert np.abs(On.resize_cubic(src, h, w) - ref).max() < 1e-6        # ATen's bicubic: the same kernel (A = -0.75) and border rule

    def same(a, b, level):
        """a, b: float fields; `level`: what one 8-bit level is worth in them (0: no 8-bit stage)."""
        d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
        tol = 2e-5 * max(1.0, float(np.abs(b).max()))
        off = d > tol
        assert off.mean() <= 0.002 and (not off.any() or d.max() <= 1.01 * level + tol), (float(off.mean()), float(d.max()), level)

    for seed in range(6):
        for exponent, px in ((-1.0, 128), (-3.0, 128), (-2.0, 5), (-2.5, 7.3), (-4.0, 2)):
            same(On.frequency_noise(np.random.RandomState(seed), h, w, exponent, px),
                 Wt.frequency_noise(np.random.RandomState(seed), h, w, exponent, px), 1.0 / 255.0)
        for o_make, p_make in ((On.fog_layers, Wt.fog_layers), (On.clouds_layers, Wt.clouds_layers)):
            got, want = o_make(np.random.RandomState(seed), h, w), p_make(np.random.RandomState(seed), h, w)
            assert len(got) == len(want)
            for (alpha, intensity), layer in zip(got, want):
                same(alpha, layer[0], 1.0 / 255.0)
                same(intensity, layer[1], 255.0 / 5.0 / 255.0 * 2.0)           # the fine field: +- mean / 5 through an 8-bit image
        got, want = On.snowflake_layers(np.random.RandomState(seed), h, w), Wt.snowflake_layers(np.random.RandomState(seed), h, w)
        assert len(got) == len(want)
        for (add, floor_), layer in zip(got, want):
            # one level of the noise image, through the gamma table's steepest step and the re-gain (<= 6) and speed (<= 2) factors
            same(add, layer[0], 12.0)
            same(floor_, layer[1], 24.0)
        got, want = On.rain_layers(np.random.RandomState(seed), h, w), Wt.rain_layers(np.random.RandomState(seed), h, w)
        assert len(got) == len(want)
        for (alpha, colour), layer in zip(got, want):
            same(alpha, layer[0], 1.0 / 255.0)
            if (alpha.reshape(-1)[:1000] == layer[0].reshape(-1)[:1000]).all():      # the drop colour is a function of those values
                assert np.abs(colour - layer[1]).max() < 1e-3
        sx, sy = On.piecewise_affine_map(np.random.RandomState(seed), h, w)
        pm = Wt.piecewise_affine_map(np.random.RandomState(seed), h, w)[0]
        assert np.abs(sx - pm[0]).max() < 1e-4 and np.abs(sy - pm[1]).max() < 1e-4
