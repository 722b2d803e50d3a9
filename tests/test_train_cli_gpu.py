"""The drop-in CLI surface on the GPU: `python train.py --config <reference-shaped YAML>` trains on synthetic batches,
writes the reference's checkpoint layout, and a second invocation resumes from it."""
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_train_cli_runs_checkpoints_and_resumes(tmp_path):
    src = open(os.path.join(REPO, "Dino", "configs", "CCD_pretrain_ViT_Tiny.yaml")).read()
    cfg = (src.replace("scheme: selfsupervised_kmeans", "scheme: synthetic\n  synthetic_samples: 256")
              .replace("imgnet_based: 1000000", "imgnet_based: 128")          # pseudo-epoch boundary every 2 iterations
              .replace("training: {epochs: 3,", "training: {epochs: 2,")
              .replace("show_iters: 200", "show_iters: 2")
              .replace("name: pre_tiny_65536", "name: cli_smoke"))
    path = tmp_path / "cli_smoke.yaml"
    path.write_text(cfg)
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29641", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(REPO, "train.py"), "--config", str(path)]
    first = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert first.returncode == 0, first.stdout[-2000:] + first.stderr[-2000:]
    assert "Starting DINO training" in first.stdout and "Training time" in first.stdout
    ckpt = tmp_path / "saved_models" / "cli_smoke" / "checkpoint.pth"
    assert ckpt.is_file(), first.stdout[-1500:]
    sd = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert {"student", "teacher", "optimizer", "epoch", "iteration", "dino_loss"} <= set(sd)
    assert all(k.startswith("module.") for k in sd["student"])              # DDP-prefixed, as the finetune script expects
    assert sd["iteration"] > 0
    second = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert second.returncode == 0, second.stdout[-2000:] + second.stderr[-2000:]
    assert "Found checkpoint" in second.stdout and f"continue to train:{sd['iteration']}" in second.stdout


@pytest.mark.gpu
def test_finetune_cli_loads_pretrain_checkpoint_trains_and_resumes(tmp_path):
    """train.py (pretraining) -> checkpoint.pth -> train_finetune.py picks the TEACHER backbone up by `module.`-prefixed
    names (train_finetune.py:190-198), trains on synthetic labelled batches, saves {net, optimizer, iteration}, resumes."""
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29642", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    pre = (open(os.path.join(REPO, "Dino", "configs", "CCD_pretrain_ViT_Tiny.yaml")).read()
           .replace("scheme: selfsupervised_kmeans", "scheme: synthetic\n  synthetic_samples: 128")
           .replace("imgnet_based: 1000000", "imgnet_based: 64")
           .replace("training: {epochs: 3,", "training: {epochs: 1,")
           .replace("name: pre_tiny_65536", "name: pre_cli"))
    (tmp_path / "pre.yaml").write_text(pre)
    r = subprocess.run([sys.executable, os.path.join(REPO, "train.py"), "--config", str(tmp_path / "pre.yaml")],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    pre_ckpt = tmp_path / "saved_models" / "pre_cli" / "checkpoint.pth"
    assert pre_ckpt.is_file()
    ft_src = open(os.path.join(REPO, "Dino", "configs", "CCD_vision_model_ARD.yaml")).read()
    ft_cfg = (ft_src.replace("scheme: supervised", "scheme: synthetic\n  synthetic_samples: 96")
              .replace("batch_size: 288}", "batch_size: 32}")
              .replace("training: {epochs: 35,", "training: {epochs: 2,")
              .replace("show_iters: 1000, eval_iters: 1000, save_iters: 100000", "show_iters: 2, eval_iters: 4, save_iters: 4")
              .replace("arch: 'vit_small'", "arch: 'vit_tiny'")
              .replace("'./saved_models/pre_small_65536/checkpoint.pth'", f"'{pre_ckpt}'")
              .replace("name: finetune_small_65536", "name: ft_cli"))
    (tmp_path / "ft.yaml").write_text(ft_cfg)
    cmd = [sys.executable, os.path.join(REPO, "train_finetune.py"), "--config", str(tmp_path / "ft.yaml")]
    first = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert first.returncode == 0, first.stdout[-2000:] + first.stderr[-2000:]
    log = first.stdout + first.stderr
    assert "Read pretrain vision model" in log and "train loss" in log and "word accuracy" in log
    ck = tmp_path / "saved_models" / "ft_cli" / "4.pth"
    assert ck.is_file(), log[-1500:]
    sd = torch.load(ck, map_location="cpu", weights_only=False)
    assert set(sd) == {"net", "optimizer", "iteration"} and sd["iteration"] == 4
    assert all(k.startswith("module.") for k in sd["net"]) and "module.decoder.classifier.weight" in sd["net"]
    # the backbone really came from the pretraining run's teacher (then trained for 4 iterations: close, not equal)
    teacher = torch.load(pre_ckpt, map_location="cpu", weights_only=False)["teacher"]
    w0, w1 = teacher["module.backbone.pos_embed"], sd["net"]["module.backbone.pos_embed"]
    assert (w0 - w1).abs().max() < 5e-3 and not torch.equal(w0, w1)
    (tmp_path / "ft2.yaml").write_text(ft_cfg.replace("checkpoint: ~", f"checkpoint: '{ck}'"))
    second = subprocess.run(cmd[:-1] + [str(tmp_path / "ft2.yaml")], cwd=tmp_path, env=env, capture_output=True, text=True,
                            timeout=600)
    assert second.returncode == 0, second.stdout[-2000:] + second.stderr[-2000:]
    assert "continue to train:4" in second.stdout
