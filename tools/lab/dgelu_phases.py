"""Per-phase cycle totals of gemm256.h's gelu'(u) product (wave 0 of every workgroup): needs a library built with -DCCD_GEMM_LAB
(CCD_EXTRA_FLAGS=-DCCD_GEMM_LAB CCD_OUT=... ccd_amd/csrc/build.sh, CCD_LIB=that file)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ccd_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
M, N, K = 131072, 1536, 384
a = torch.randn(M, K, device=dev).to(BF); b = (torch.randn(N, K, device=dev) * 0.1).to(BF)
aux = torch.randn(M, N, device=dev).to(BF); out = torch.empty(M, N, device=dev, dtype=BF); c2 = torch.empty(M, N, device=dev, dtype=BF)
stamps = torch.zeros(8 * 1024 * 2, device=dev)
fn = lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux=aux, out=out, out2=c2, m_fastest=64, colsum=stamps)
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
print("ms per launch", e0.elapsed_time(e1) / 10)
st = stamps.view(torch.int64).view(-1, 8)[:256].double().cpu()
names = ["first-tile wait", "compute+dma issue", "vmcnt wait", "k barrier", "epi stage", "epi barriers", "epi rows", "epi aux wait"]
tot = st.sum(1).mean().item()
for i, n in enumerate(names):
    print(f"  {n:18s} {st[:, i].mean().item():12.0f}  {100 * st[:, i].mean().item() / tot:5.1f} %")
print("  total (s_memtime ticks, 100 MHz)", tot)
