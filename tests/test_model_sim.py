"""The whole product stack (modules, engine, arena, fused optimizer) on the reference's tiny-model fixture, with the
HIP kernels executed by the CPU SIMT executor."""
import pytest
import torch

from backends import Backend
import model_checks as mc


@pytest.fixture(scope="module")
def sim():
    with Backend("sim") as b:
        yield b


def test_tiny_training_iteration_sim(sim):
    mc.check_tiny_step(sim.device)


@pytest.mark.parametrize("lnbwd", [True, False])
def test_tiny_training_iteration_rotating_gb_buffers_sim(sim, lnbwd):
    """The backward pass of the side-stream mode (two gb buffers used in turn; LayerNorm backward fused into the data-gradient
    product or as its own launch behind it) must give the same iteration - here without a second stream, on the CPU executor."""
    from ccd_amd import engine
    saved = (engine.Fusion.double_gb, engine.Fusion.lnbwd)
    engine.Fusion.double_gb, engine.Fusion.lnbwd = True, lnbwd
    try:
        mc.check_tiny_step(sim.device)
    finally:
        engine.Fusion.double_gb, engine.Fusion.lnbwd = saved


def test_wide_768_model_iteration_sim(sim):
    """E = 768 / 12 heads (BASELINE config #4's shape) on the CPU executor: a 3-block model of that width, one full pretraining
    iteration against the CPU oracle - the unfused GEMM path, LayerNorm kernels beyond 512 columns, 12-head attention."""
    mc.check_pretrain_arch_vs_oracle(sim.device, arch=None, B=1, out_dim=256, dims=(768, 3, 12, (1, 2, 3)))


def test_optimizer_steps_sim(sim):
    mc.check_optimizer_host_runs_ahead(sim.device, steps=3)


def test_checkpoint_resume_sim(sim, tmp_path):
    mc.check_checkpoint_resume(sim.device, tmp_path)


def test_finetune_against_oracle_sim(sim):
    """DINO_Finetune (2-block backbone, 2 decoder layers) - loss, gradients, AdamW, greedy decoding."""
    mc.check_finetune_against_oracle(sim.device, B=3, max_seq_len=25, steps=2)


def test_finetune_dropout_sim(sim):
    mc.check_finetune_dropout(sim.device)


def test_finetune_fused_mlp_sim(sim, monkeypatch):
    """A backbone whose width is a multiple of 128 takes the fused MLP kernel in both passes (forward keeps only the
    pre-activation, backward re-derives gelu(u)): the whole iteration against the oracle, gradients included."""
    from ccd_amd import ops
    calls = []
    real = ops.mlp_fused
    monkeypatch.setattr(ops, "mlp_fused", lambda *a, **k: (calls.append(k.get("store_u")), real(*a, **k))[1])
    _run_fused(sim)
    assert calls and any(calls), "the fused MLP kernel was not on the path"


def _run_fused(sim):
    mc.check_finetune_against_oracle(sim.device, arch="vit_test128", vit_kw=dict(embed_dim=128, depth=2, heads=2), B=2,
                                     max_seq_len=25, steps=2, decode=False)


def test_training_iterations_under_late_dma_model():
    """One pretraining and one finetune iteration with the executor delivering LDS-DMA data as late as the hardware may
    (CCD_SIM_DMA=late, see tests/test_kernels_sim.py::test_kernels_under_late_dma_model): every default kernel's counted waits
    inside whole forward / backward passes."""
    import os
    import subprocess
    import sys
    if os.environ.get("CCD_SIM_DMA", "").startswith("l"):
        return
    env = dict(os.environ, CCD_SIM_DMA="late")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider", "-k",
                        "tiny_training_iteration or finetune_against_oracle"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
