#!/bin/bash
# round 6, job 7: where the step stands - steady-state kernel table + launch sequence, bench line, N > 1 path on one rank
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
bash tools/prof_bench.sh r06_bench_b256 > gpurun_out/r06_prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline 2> gpurun_out/r06_j7_a.err | tail -1 > gpurun_out/r06_j7_a.json
BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline 2> gpurun_out/r06_j7_dist.err | tail -1 > gpurun_out/r06_j7_dist.json
python bench.py --no-cpu-baseline --batch 64 2> gpurun_out/r06_j7_b64.err | tail -1 > gpurun_out/r06_j7_b64.json
for f in r06_j7_a r06_j7_dist r06_j7_b64; do python - gpurun_out/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"))
PY
done
head -60 gpurun_out/r06_bench_b256_steady_state.md
