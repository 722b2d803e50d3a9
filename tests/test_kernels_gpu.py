"""Parity of every HIP kernel on a real MI355X, through the C ABI of libccd_hip.so (run with -m gpu)."""
import pytest
import torch

from backends import Backend
import kernel_checks as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    with Backend("hip") as b:
        yield b


@pytest.mark.parametrize("M,N,K", [(200, 136, 128), (1024, 1152, 384), (4096, 384, 1536), (96, 65536, 256),
                                   (1000, 192, 192)])
def test_gemm_nt(hip, M, N, K):
    kc.check_gemm_nt(hip.device, M, N, K)


@pytest.mark.parametrize("Mc,P,Q,splits", [(300, 136, 72, 3), (4096, 1152, 384, 0), (8192, 384, 1536, 0),
                                           (96, 65536, 256, 1), (64, 8, 264, 0)])
def test_gemm_tn(hip, Mc, P, Q, splits):
    kc.check_gemm_tn(hip.device, Mc, P, Q, splits=splits)


@pytest.mark.parametrize("rows,E", [(37, 192), (4096, 384), (1001, 512)])
def test_layernorm(hip, rows, E):
    kc.check_layernorm(hip.device, rows, E)


@pytest.mark.parametrize("views,heads,spike", [(1, 2, False), (16, 6, False), (3, 8, True)])
def test_attention(hip, views, heads, spike):
    kc.check_attention(hip.device, views, heads, spike=spike)
