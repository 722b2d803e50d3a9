"""SURVEY 8(f) row 4 on the GPU: the evaluation entry point (test.py / TextAccuracy.compute) reproduces the predictions and
the metric dictionary the REAL reference produced for the seeded recogniser, from tensors and from an LMDB benchmark
through the CLI; checkpoints in the published layouts are rebuilt into running networks and exported back."""
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from backends import Backend

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    with Backend("hip") as b:
        yield b


def _stat(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), t.pow(2).sum().sqrt().item()])


def test_text_accuracy_compute_matches_reference_run(hip):
    from ccd_amd import finetune as ft
    from ccd_amd.metric.eval_acc import TextAccuracy
    from ccd_amd.parallel import DataParallel
    g = np.load(os.path.join(GOLD, "eval_acc.npz"))
    torch.manual_seed(0)
    model = ft.build_model(ft.FinetuneConfig(arch="vit_tiny", drop_path_rate=0.0, decoder_n_layers=2), hip.device, dropout=0.0)
    model.eval()
    gen = torch.Generator().manual_seed(4321)
    batches = [torch.randn(6, 3, 32, 128, generator=gen) for _ in range(3)]
    np.testing.assert_allclose(np.stack([_stat(b) for b in batches]), g["image_stat"], rtol=1e-12)
    gts = [tuple(str(s) for s in g["gt"][6 * b: 6 * b + 6]) for b in range(3)]
    loader = [(batches[b], [gts[b]]) for b in range(3)]             # what default_collate makes of (image, [label]) samples
    metric = TextAccuracy(charset_path=None, case_sensitive=False, model_eval="vision")
    res = metric.compute(DataParallel(model), loader)
    want = dict(zip([str(n) for n in g["names"]], g["values"]))
    # the bf16 recogniser must decode the same strings as the fp32 reference ...
    with torch.no_grad():
        idx, _ = model.label_convertor.tensor2idx(model(batches[0].to(hip.device), None, return_loss=False))
    assert model.label_convertor.idx2str(idx) == [str(s) for s in g["pred"][:6]]
    # ... hence the same scores
    for k in ("ccr", "cwr", "ted", "ned", "ted/w", "words"):
        assert res[k] == pytest.approx(float(want[k]), rel=1e-9), k
    assert res["time"] > 0


def test_published_layouts_rebuild_and_export(hip, tmp_path):
    from ccd_amd import checkpoint
    keys = json.load(open(os.path.join(GOLD, "state_keys.json")))["vit_tiny"]
    gen = torch.Generator().manual_seed(3)
    fab = lambda table: {"module." + k: (torch.randn(shape, generator=gen) * 0.02 if dt == "torch.float32"
                                         else torch.tensor(3, dtype=torch.int64)) for k, shape, dt in table}
    ckpt = {"student": fab(keys["student"]), "teacher": fab(keys["teacher"]), "epoch": 1, "iteration": 10}
    torch.save(ckpt, tmp_path / "pre.pth")
    loaded = torch.load(tmp_path / "pre.pth", map_location="cpu", weights_only=False)
    student, teacher = checkpoint.build_pretrain_models(loaded, device=hip.device)
    checkpoint.export_pretrain(student, teacher, tmp_path / "out.pth", epoch=1, iteration=10)
    back = torch.load(tmp_path / "out.pth", map_location="cpu", weights_only=False)
    for name in ("student", "teacher"):
        assert list(back[name]) == list(loaded[name])
        assert all(torch.equal(back[name][k], loaded[name][k]) for k in loaded[name]), name
    # the finetune layout: train_finetune.py's own checkpoint -> rebuilt -> same predictions
    from ccd_amd import finetune as ft
    torch.manual_seed(5)
    model = ft.build_model(ft.FinetuneConfig(arch="vit_tiny", drop_path_rate=0.0, decoder_n_layers=2), hip.device, dropout=0.0)
    from ccd_amd.parallel import DataParallel
    checkpoint.export_finetune(DataParallel(model), tmp_path / "ft.pth", iteration=3)
    ft_ckpt = torch.load(tmp_path / "ft.pth", map_location="cpu", weights_only=False)
    assert checkpoint.describe(ft_ckpt)["net"]["decoder_layers"] == 2
    rebuilt = checkpoint.build_finetune_model(ft_ckpt, ft.FinetuneConfig(arch="vit_tiny", drop_path_rate=0.0, decoder_n_layers=2),
                                              device=hip.device)
    img = torch.randn(4, 3, 32, 128, generator=gen).to(hip.device)
    model.eval(); rebuilt.module.eval()
    with torch.no_grad():
        torch.testing.assert_close(rebuilt(img, None, return_loss=False), model(img, None, return_loss=False), rtol=0, atol=0)


def _png(arr):
    from PIL import Image
    out = io.BytesIO()
    Image.fromarray(arr).save(out, format="PNG")
    return out.getvalue()


def test_eval_cli_on_lmdb_benchmarks(tmp_path):
    """`python test.py --config <reference-shaped YAML>`: a checkpoint written by train_finetune.py's layout, two benchmark
    LMDBs (one of them a folder of two), the reference's report format."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from ccd_amd.dataset import lmdb_file
    rs = np.random.RandomState(0)
    words = ["hello", "World", "MI355X", "text", "ccd", "a1b2", "Hip-Kernel", "x"]
    roots = []
    for name, n in (("bench_a", 10), ("bench_b/part1", 6), ("bench_b/part2", 5)):
        recs = {b"num-samples": str(n).encode()}
        for i in range(1, n + 1):
            recs[b"image-%09d" % i] = _png(rs.randint(0, 256, size=(int(rs.randint(20, 50)), int(rs.randint(60, 200)), 3)).astype(np.uint8))
            recs[b"label-%09d" % i] = words[i % len(words)].encode()
        lmdb_file.write_lmdb(str(tmp_path / "eval" / name), recs)
    roots = [str(tmp_path / "eval" / "bench_a"), str(tmp_path / "eval" / "bench_b")]
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), LOCAL_RANK="0")
    mk = ("import sys, torch; sys.path.insert(0, %r)\n"
          "from ccd_amd import finetune as ft, checkpoint\n"
          "from ccd_amd.parallel import DataParallel\n"
          "from Dino.utils.utils import Config\n"
          "torch.manual_seed(0)\n"
          "m = ft.build_model(Config(%r), 'cuda')\n"
          "checkpoint.export_finetune(DataParallel(m), %r, iteration=1)\n")
    src = open(os.path.join(REPO, "Dino", "configs", "CCD_vision_model_ARD.yaml")).read()
    import re
    cfg = re.sub(r"test: \{roots: \[[^\]]*\]", "test: {roots: %s" % json.dumps(roots).replace('"', "'"), src, flags=re.S)
    cfg = cfg.replace("num_workers: 8", "num_workers: 2")
    (tmp_path / "eval.yaml").write_text(cfg)
    r = subprocess.run([sys.executable, "-c", mk % (REPO, str(tmp_path / "eval.yaml"), str(tmp_path / "ft.pth"))], cwd=tmp_path,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    run = subprocess.run([sys.executable, os.path.join(REPO, "test.py"), "--config", str(tmp_path / "eval.yaml"), "--checkpoint",
                          str(tmp_path / "ft.pth"), "--batch_size", "4"], cwd=tmp_path, env=env, capture_output=True, text=True,
                         timeout=900)
    assert run.returncode == 0, run.stdout[-2500:] + run.stderr[-2500:]
    assert "dataset: IIIT5k_3000 --> word_num: 10.0 --> accuracy: " in run.stdout
    assert "dataset: SVT --> word_num: 11.0 --> accuracy: " in run.stdout          # the folder of two LMDBs is one benchmark
    assert "total_accuracy: " in run.stdout
