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
