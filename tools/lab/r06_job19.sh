#!/bin/bash
# round 6, job 19: bf16 gradient stream at E = 768 (stand-alone LayerNorm-backward passes) - parity test of the 768 / 12 shape against the
# reference fixture, same-box bench A/B (CCD_G_BF16 = 0 / default)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "other_archs" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -4
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); k=d["roofline"]["by_kind_ms_per_step"]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], {n: k.get(n) for n in ("layernorm_bwd", "layernorm_fwd")})
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j19_768_g16_a --arch vit_base_768 --batch 128
ENVV=(CCD_G_BF16=0); run r06_j19_768_g32_a --arch vit_base_768 --batch 128
ENVV=(X=1); run r06_j19_768_g16_b --arch vit_base_768 --batch 128
ENVV=(CCD_G_BF16=0); run r06_j19_768_g32_b --arch vit_base_768 --batch 128
