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


def test_checkpoint_resume_sim(sim, tmp_path):
    mc.check_checkpoint_resume(sim.device, tmp_path)


def test_finetune_against_oracle_sim(sim):
    """DINO_Finetune (2-block backbone, 2 decoder layers) - loss, gradients, AdamW, greedy decoding."""
    mc.check_finetune_against_oracle(sim.device, B=3, max_seq_len=25, steps=2)


def test_finetune_dropout_sim(sim):
    mc.check_finetune_dropout(sim.device)
