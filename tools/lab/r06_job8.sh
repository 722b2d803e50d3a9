#!/bin/bash
# round 6, job 8: the N > 1 path on one rank after the reserve-window changes (same box A/B against the plain step)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j8_plain_a
ENVV=(BENCH_FORCE_DIST=1); run r06_j8_dist_a
ENVV=(X=1); run r06_j8_plain_b
ENVV=(BENCH_FORCE_DIST=1); run r06_j8_dist_b
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "distributed or no_grad" 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -3
