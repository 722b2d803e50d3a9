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


def test_g_bf16_matches_fp32_sim(sim):
    """The bf16 residual-gradient stream (ccd_*_g16) inside a training iteration against the fp32 stream, on the CPU executor."""
    mc.check_g_bf16_matches_fp32(sim.device, E=128, batch=2)


def test_mlp_bwd_fused_matches_two_launches_sim(sim, monkeypatch):
    """The one-launch MLP backward (mlp_bwd.h) and the gelu(u)-storing forward block half inside a training iteration, CPU executor."""
    monkeypatch.setenv("CCD_SIM_CUS", "4")
    mc.check_mlp_bwd_fused_matches_two_launches(sim.device, E=256, batch=2, depth=3)


def test_no_grad_train_droppath_sim(sim):
    """train() + no_grad + drop_path > 0: blocks with a DropPath mask fall back from ccd_proj_mlp_fused."""
    mc.check_no_grad_train_droppath(sim.device, E=128, views=2)


def test_head_loss_fusion_matches_unfused_sim(sim, monkeypatch):
    """ccd_head_loss_fwd / _bwd inside the training iteration (LazyLogits) against the chain it replaces, on the CPU executor."""
    monkeypatch.setenv("CCD_SIM_CUS", "8")
    mc.check_head_loss_fusion_matches_unfused(sim.device, batch=2)


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
    from ccd_amd import engine, ops
    calls, pcalls = [], []
    real, preal = ops.mlp_fused, ops.proj_mlp_fused
    monkeypatch.setattr(ops, "mlp_fused", lambda *a, **k: (calls.append(k.get("store_u")), real(*a, **k))[1])
    monkeypatch.setattr(ops, "proj_mlp_fused", lambda *a, **k: (pcalls.append(k.get("save")), preal(*a, **k))[1])
    _run_fused(sim)      # default: proj + residual + LayerNorm-2 ride in front of the fused MLP (ccd_proj_mlp_fused)
    assert pcalls and any(pcalls) and not calls, "the fused block-half kernel was not on the path"
    monkeypatch.setattr(engine.Fusion, "proj_mlp", False)
    _run_fused(sim)      # the two separate launches it replaces
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


def test_graphed_training_step_bookkeeping_sim(sim, monkeypatch):
    """pretrain.GraphedTrainingStep's host side on the CPU executor: there is no HIP graph here, so `_record` is replaced by a
    stand-in whose replay() re-issues the captured launches with the launch-time constants of the capture (epoch / teacher
    temperature closed over, the DropPath seed counter rewound to its value at capture) - what a graph replay does.  Everything
    that must reach the kernels through device memory (momentum, seed offset, the AdamW table, the batch) then has to be right
    for the run to agree with the eager one; re-captures are counted."""
    from ccd_amd import engine, pretrain

    def record(self, body):
        seeds = engine._DROPPATH_SEED
        at_capture, d_seed = seeds["calls"], seeds["device"]

        class Replay:
            def replay(_):
                now = seeds["calls"]
                seeds["calls"] = at_capture
                engine.set_device_droppath_seed(d_seed)
                try:
                    body()
                finally:
                    engine.set_device_droppath_seed(None)
                    seeds["calls"] = now
        return Replay(), 1

    masks, ema_args = [], []
    real_scales, real_ema = engine.ops.droppath_scales, engine.ops.ema

    def spy_scales(*a, **kw):
        out = real_scales(*a, **kw)
        masks.append(out.clone())
        return out

    def spy_ema(teacher, student, mirror, m, d_m=None):
        ema_args.append(float(m) if d_m is None else float(d_m[0]))
        return real_ema(teacher, student, mirror, m, d_m)

    monkeypatch.setattr(engine.ops, "droppath_scales", spy_scales)
    monkeypatch.setattr(engine.ops, "ema", spy_ema)
    monkeypatch.setattr(pretrain.GraphedTrainingStep, "_record", record)
    steps = 5
    mc.check_graphed_step_matches_eager(sim.device, steps=steps, B=1, drop_path_rate=0.5)
    # exactly the eager run's DropPath masks and momenta reached the kernels of the replays
    assert len(masks) == 2 * steps and all(torch.equal(masks[i], masks[steps + i]) for i in range(steps))
    assert len({m.numpy().tobytes() for m in masks[steps:]}) >= 3          # and they change from replay to replay
    per = len(ema_args) // (2 * steps)
    assert ema_args[:per * steps] == pytest.approx(ema_args[per * steps:], abs=1e-7), ema_args
