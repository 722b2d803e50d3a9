#!/bin/bash
# round 6, job 10: do the 6-us gaps around the fused block half (the one kernel with ~300 B of scratch) answer to ROCr's scratch settings?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); k=d["roofline"]["by_kind_ms_per_step"]; print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], k.get("proj_mlp_fused"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
ENVV=(X=1); run r06_j10_plain_a
ENVV=(HSA_SCRATCH_SINGLE_LIMIT=4000000000); run r06_j10_limit_a
ENVV=(HSA_NO_SCRATCH_RECLAIM=1); run r06_j10_noreclaim_a
ENVV=(HSA_NO_SCRATCH_RECLAIM=1 HSA_SCRATCH_SINGLE_LIMIT=4000000000 HSA_NO_SCRATCH_THREAD_LIMITER=1); run r06_j10_all_a
ENVV=(X=1); run r06_j10_plain_b
