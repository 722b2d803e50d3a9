"""SURVEY 8(f) row 4 on the host: TextAccuracy's arithmetic against the numbers the REAL reference class produced
(tests/golden/eval_acc.npz, tools/gen_golden.py `gen_eval`), checkpoint description / conversion of both published layouts."""
import json
import os

import numpy as np
import pytest
import torch

from ccd_amd import checkpoint
from ccd_amd.metric.eval_acc import TextAccuracy, levenshtein

HERE = os.path.dirname(os.path.abspath(__file__))


def test_text_accuracy_matches_reference_numbers():
    g = np.load(os.path.join(HERE, "golden", "eval_acc.npz"))
    pred, gt = [str(s) for s in g["pred"]], [str(s) for s in g["gt"]]
    metric = TextAccuracy(charset_path=None, case_sensitive=False, model_eval="vision")
    for b in range(0, len(pred), 6):                               # the reference scored three batches of six
        metric.update(gt[b:b + 6], pred[b:b + 6])
    res = metric.result()
    want = dict(zip([str(n) for n in g["names"]], g["values"]))
    assert list(res) == ["ccr", "cwr", "ted", "ned", "ted/w", "words", "time"]
    for k in ("ccr", "cwr", "ted", "ned", "ted/w", "words"):
        assert res[k] == pytest.approx(float(want[k]), rel=1e-12, abs=0), k
    assert res["cwr"] == pytest.approx(12 / 18)                    # exact + case + punctuation variants count as correct words


def test_levenshtein_and_normalisation():
    assert levenshtein("kitten", "sitting") == 3 and levenshtein("", "abc") == 3 and levenshtein("flaw", "lawn") == 2
    m = TextAccuracy(None, False, "vision")
    m.update(["Hello", "W-or1d", "a^b", "中文a"], ["hello", "word", "a^b", "中文A"])
    r = m.result()
    assert r["cwr"] == 0.75 and r["ted"] == 1.0 and r["words"] == 4
    assert r["ned"] == pytest.approx(1.0 / 6)                       # divided by the RAW ground-truth length (6 for 'W-or1d')
    with pytest.raises(NotImplementedError):
        TextAccuracy(None, True, "vision").update(["a"], ["a"])


def _fabricate(table, gen):
    return {"module." + k: (torch.randn(shape, generator=gen) if dt == "torch.float32" else torch.tensor(3, dtype=torch.int64))
            for k, shape, dt in table}


@pytest.mark.parametrize("arch,width", [("vit_tiny", 192), ("vit_small", 384), ("vit_base", 512)])
def test_describe_and_convert_pretrain_checkpoint(tmp_path, arch, width):
    keys = json.load(open(os.path.join(HERE, "golden", "state_keys.json")))[arch]
    gen = torch.Generator().manual_seed(1)
    ckpt = {"student": _fabricate(keys["student"], gen), "teacher": _fabricate(keys["teacher"], gen), "optimizer": {}, "epoch": 2,
            "iteration": 99, "dino_loss": {"center": torch.zeros(1, 8)}}
    path = tmp_path / "checkpoint.pth"
    torch.save(ckpt, path)
    info = checkpoint.describe(str(path))
    assert info["kind"] == "pretrain" and info["epoch"] == 2 and info["iteration"] == 99
    assert info["student"]["arch"] == arch and info["student"]["embed_dim"] == width and info["student"]["depth"] == 12
    assert info["student"]["patch_size"] == 4 and info["student"]["has_segmentation"] and not info["teacher"]["has_segmentation"]
    assert info["student"]["ddp_prefixed"] and info["teacher"]["tensors"] < info["student"]["tensors"] == 252
    assert info["student"]["out_dim"] == ckpt["student"]["module.head.last_layer.weight_v"].shape[0]
    checkpoint.main(["convert", str(path), str(tmp_path / "teacher.pth"), "--strip-prefix", "--only", "teacher"])
    bare = torch.load(tmp_path / "teacher.pth", map_location="cpu", weights_only=False)
    assert all(not k.startswith("module.") for k in bare) and len(bare) == len(ckpt["teacher"])
    assert torch.equal(bare["backbone.pos_embed"], ckpt["teacher"]["module.backbone.pos_embed"])
    checkpoint.convert(str(path), str(tmp_path / "same.pth"))
    again = torch.load(tmp_path / "same.pth", map_location="cpu", weights_only=False)
    assert list(again["student"]) == list(ckpt["student"]) and again["iteration"] == 99
    with pytest.raises(ValueError):
        checkpoint.describe({"model": {}})


def test_describe_finetune_checkpoint():
    sd = {"module.backbone.pos_embed": torch.zeros(1, 257, 384), "module.backbone.patch_embed.proj.weight": torch.zeros(384, 3, 4, 4),
          "module.backbone.blocks.11.norm1.weight": torch.zeros(384), "module.decoder.layer_stack.5.norm1.weight": torch.zeros(512)}
    info = checkpoint.describe({"net": sd, "optimizer": {}, "iteration": 7})
    assert info["kind"] == "finetune" and info["net"]["arch"] == "vit_small" and info["net"]["decoder_layers"] == 6
    assert info["net"]["depth"] == 12 and info["iteration"] == 7
