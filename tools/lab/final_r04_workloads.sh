#!/bin/bash
# round 4: the other workloads of the bench (one JSON line each under gpurun_out/) + the stored-gelu A/B on the final kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
ENVV=(X=1)
run r04_bench_epoch30 --epoch 30
run r04_bench_b64 --batch 64
run r04_bench_vit_base_b128 --arch vit_base --batch 128
run r04_bench_vit_base_768_b128 --arch vit_base_768 --batch 128
run r04_bench_finetune_b512 --workload finetune --batch 512
run r04_bench_b256_again
ENVV=(CCD_STORE_GACT=1)
run r04_bench_b256_store_gact
ENVV=(X=1)
run r04_bench_b256_again2
ENVV=(CCD_STORE_GACT=1)
run r04_bench_b256_store_gact2
