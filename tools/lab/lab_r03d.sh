#!/bin/bash
# round 3, GPU batch d: the whole GPU suite, attention-backward chunking (MALL reuse), benches of every workload
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03d_gputests.log
ATTN_ONLY=1 python tools/rowproj_lab.py > gpurun_out/r03d_attn_lab.jsonl 2> gpurun_out/r03d_attn_lab.err
run() { name=$1; shift; env "$@" > gpurun_out/r03d_bench_$name.json 2> gpurun_out/r03d_bench_$name.err; }
run chunks1 CCD_ATTN_CHUNKS=1 python bench.py --no-cpu-baseline
run chunks2 CCD_ATTN_CHUNKS=2 python bench.py --no-cpu-baseline
run chunks4 CCD_ATTN_CHUNKS=4 python bench.py --no-cpu-baseline
run vit_base_b128 CCD_X=0 python bench.py --no-cpu-baseline --arch vit_base --batch 128
run vit_base_768_b128 CCD_X=0 python bench.py --no-cpu-baseline --arch vit_base_768 --batch 128
run finetune_b512 CCD_X=0 python bench.py --no-cpu-baseline --workload finetune --batch 512
run epoch30 CCD_X=0 python bench.py --no-cpu-baseline --epoch 30
run b64 CCD_X=0 python bench.py --no-cpu-baseline --batch 64 --steps 20
cat gpurun_out/r03d_gputests.log gpurun_out/r03d_attn_lab.jsonl
for f in gpurun_out/r03d_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get("roofline",{}).get("by_kind_ms_per_step",{})
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), {x:k.get(x) for x in ("mlp_fused","gemm_nt_bf16","attention_bwd","gemm_nt_lnbwd","gemm_tn_atomic","gemm_nt_dgelu")})
except Exception as e:
    print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
