#!/bin/bash
# round 6, job 18: lab times of patch_embed_fwd / small_matmul, new library against the previous one
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
cat > /tmp/pe_lab.py <<'PY'
import sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from ccd_amd import ops
from mlp_lab import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
V, E = 512, 384
img = torch.randn(V, 3, 32, 128, generator=g).to(dev)
w = (torch.randn(E, 3, 4, 4, generator=g) * 0.1).to(dev); b = torch.zeros(E, device=dev); pos = torch.randn(256, E, generator=g).to(dev)
rs = torch.randn(256, 256, generator=g).to(dev); pe = torch.randn(256, E, generator=g).to(dev); out = torch.empty(256, E, device=dev)
print(json.dumps({"patch_embed_fwd_ms": round(timeit(lambda: ops.patch_embed_fwd(img, w, b, pos)), 4),
                  "small_matmul_ms": round(timeit(lambda: ops.small_matmul(rs, pe, out)), 4),
                  "small_matmul_transa_acc_ms": round(timeit(lambda: ops.small_matmul(rs, pe, out, trans_a=True, accumulate=True)), 4)}))
PY
for i in 1 2; do
python /tmp/pe_lab.py 2>/dev/null | sed 's/^{/{"lib": "new", /' | tee -a gpurun_out/r06_patch_embed_lab.jsonl
CCD_HIP_LIB=$PWD/ccd_amd/lab_prev.so python /tmp/pe_lab.py 2>/dev/null | sed 's/^{/{"lib": "prev", /' | tee -a gpurun_out/r06_patch_embed_lab.jsonl
done
