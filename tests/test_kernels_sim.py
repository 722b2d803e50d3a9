"""Kernel logic under the CPU SIMT executor (tests/hipsim): the product's kernel sources + C ABI compiled for the
host, tiny shapes.  Validates indexing / layouts / epilogues here, where there is no GPU."""
import pytest

from backends import Backend
import kernel_checks as kc


@pytest.fixture(scope="module")
def sim():
    with Backend("sim") as b:
        yield b


def test_gemm_nt_sim(sim):
    kc.check_gemm_nt(sim.device, M=200, N=136, K=128)


def test_gemm_tn_sim(sim):
    kc.check_gemm_tn(sim.device, Mc=300, P=136, Q=72, splits=3)
    kc.check_gemm_tn(sim.device, Mc=64, P=8, Q=264)


def test_layernorm_sim(sim):
    kc.check_layernorm(sim.device, rows=37, E=192)
    kc.check_layernorm(sim.device, rows=9, E=384)


def test_attention_sim(sim):
    kc.check_attention(sim.device, views=1, heads=2)
