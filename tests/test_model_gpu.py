"""End-to-end parity of the product stack on a real MI355X (run with -m gpu)."""
import json
import os

import pytest
import torch

from backends import Backend
import model_checks as mc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    with Backend("hip") as b:
        yield b


def test_tiny_training_iteration(hip):
    mc.check_tiny_step(hip.device, batch=8)     # the north-star tolerance (1e-3) on ~56 selected rows
    mc.check_tiny_step(hip.device, batch=2)


def test_vit_small_b8_two_iterations_vs_reference(hip):
    """BASELINE config #1: distillation loss within 1e-3 of the CPU reference, index map bit-exact."""
    report = mc.check_small_steps(hip.device, loss_tol=1e-3)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_small_step.json", "w") as f:
        json.dump(report, f)


def test_vit_small_b8_four_iterations_vs_reference_nonzero_head_biases(hip):
    """Multi-iteration parity against the real reference with no excluded tensor (see check_small3_steps)."""
    report = mc.check_small3_steps(hip.device, loss_tol=1e-3)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_small3_step.json", "w") as f:
        json.dump(report, f)


def test_distributed_path_on_one_rank(hip):
    """RCCL init, SyncBatchNorm, bucketed async all-reduces against the persistent GEMM grids: executed once on this GPU."""
    report = mc.check_dist_world1(hip.device)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/dist_world1.json", "w") as f:
        json.dump(report, f)


def test_fold_tap_matches_separate_launch(hip):
    """Taps behind blocks 1 and 2 fold into the qkv products of blocks 2 and 3; the tap behind the last block keeps its own launch."""
    report = mc.check_fold_tap_matches_separate(hip.device, E=384, batch=4)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fold_tap_vs_separate.json", "w") as f:
        json.dump(report, f)


def test_mlp_bwd_fused_matches_two_launches(hip):
    """ccd_proj_mlp_fused_gact + ccd_mlp_bwd_fused inside a training iteration against the gelu'(u) product + LayerNorm-backward product."""
    report = mc.check_mlp_bwd_fused_matches_two_launches(hip.device, E=384, batch=4)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/mlp_bwd_fused_vs_two_launches.json", "w") as f:
        json.dump(report, f)


def test_g_bf16_matches_fp32(hip):
    report = mc.check_g_bf16_matches_fp32(hip.device, E=384, batch=8)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/g_bf16_vs_fp32.json", "w") as f:
        json.dump(report, f)


def test_no_grad_train_droppath(hip):
    mc.check_no_grad_train_droppath(hip.device, E=384, views=8)


def test_head_loss_fusion_matches_unfused(hip):
    """The head's last product left to the loss (ccd_head_loss_fwd / _bwd, logits never written) against the materialised chain:
    one iteration each at out_dim = 4096, same losses / centre / gradients."""
    report = mc.check_head_loss_fusion_matches_unfused(hip.device, batch=8, out_dim=4096)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/head_loss_fusion.json", "w") as f:
        json.dump(report, f)


def test_optimizer_host_runs_ahead(hip):
    mc.check_optimizer_host_runs_ahead(hip.device)


def test_graphed_training_step_matches_eager(hip):
    mc.check_graphed_step_matches_eager(hip.device)


def test_full_batch_iteration_equals_micro_batches(hip):
    report = mc.check_full_batch_equals_micro_batches(hip.device, B=256)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/full_batch_check.json", "w") as f:
        json.dump(report, f)


def test_properties_at_full_batch(hip):
    m = mc.check_properties_full_size(hip.device, B=256)
    assert 3 * 256 < m < 10 * 256


def test_smoke_entry(hip):
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize("tag", ["tiny", "small"])
def test_finetune_vs_reference(hip, tag):
    """DINO_Finetune against the real reference's recorded iterations (vit_tiny/2 layers, vit_small/6)."""
    report = {}
    try:
        mc.check_finetune_golden(hip.device, tag, report=report)
    finally:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/parity_finetune_{tag}.json", "w") as f:
            json.dump(report, f)


def test_finetune_vs_oracle(hip):
    from oracle import ccd_oracle as O
    mc.check_finetune_against_oracle(hip.device, arch="vit_small", vit_kw=O.ARCH["vit_small"], n_layers=6, B=8, steps=1)


def test_finetune_dropout(hip):
    mc.check_finetune_dropout(hip.device, B=8, n_layers=2)


def test_greedy_decoding_hip_graph_matches_eager(hip, monkeypatch):
    """forward_test replays the 25 decoding steps from a captured HIP graph: identical output to issuing the kernels one by
    one, also after the weights changed (the graph reads the arena in place) and for a second batch shape."""
    from ccd_amd import finetune as ft
    torch.manual_seed(2)
    model = ft.build_model(ft.FinetuneConfig(arch="vit_tiny", drop_path_rate=0.0, decoder_n_layers=2), hip.device)
    model.eval()
    g = torch.Generator().manual_seed(5)
    for B in (16, 16, 5):
        img = torch.randn(B, 3, 32, 128, generator=g).to(hip.device)
        with torch.no_grad():
            monkeypatch.setenv("CCD_DECODE_GRAPH", "1")
            graphed = model(img, None, return_loss=False)
            monkeypatch.setenv("CCD_DECODE_GRAPH", "0")
            eager = model(img, None, return_loss=False)
        assert graphed.shape == (B, 25, 92) and torch.equal(graphed, eager)
        with torch.no_grad():                                   # perturb the weights in place between evaluations
            model.arena.flat.mul_(1.01)
        model.arena.refresh_mirrors()
    assert len(model.decoder._graphs) == 2


@pytest.mark.parametrize("arch", ["vit_base", "vit_tiny", "vit_base_768"])
def test_other_archs_pretrain_iteration_vs_oracle(hip, arch):
    """BASELINE config #4's architecture (vit_base: E 512 / 8 heads, unfused residual + LayerNorm path; vit_base_768: the
    768 / 12 shape that config's text names) and vit_tiny."""
    rep = mc.check_pretrain_arch_vs_oracle(hip.device, arch, B=4)
    print(arch, rep)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)      # kept: the deltas are part of the parity record (profiles/)
    with open(os.path.join(ROOT, "gpurun_out", f"parity_arch_{arch}.json"), "w") as f:
        json.dump(rep, f, indent=1)


def test_finetune_properties_at_full_batch(hip):
    mc.check_finetune_properties_full_size(hip.device)
