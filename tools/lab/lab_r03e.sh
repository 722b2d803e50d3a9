#!/bin/bash
# round 3, GPU batch e: fused MLP backward (mlp_bwd.h) - tests, A/B in the step, kernel trace
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r03e_gputests.log
run() { name=$1; shift; env "$@" > gpurun_out/r03e_bench_$name.json 2> gpurun_out/r03e_bench_$name.err; }
run mlpbwd1 CCD_FUSE_MLP_BWD=1 python bench.py --no-cpu-baseline
run mlpbwd0 CCD_FUSE_MLP_BWD=0 python bench.py --no-cpu-baseline
run mlpbwd1_again CCD_FUSE_MLP_BWD=1 python bench.py --no-cpu-baseline
run finetune_b512 CCD_X=0 python bench.py --no-cpu-baseline --workload finetune --batch 512
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03e -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timer --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r03e_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_db_summary.py gpurun_out/prof_r03e/bench_results.db --steps 2 > gpurun_out/r03e_steady_state.md 2> gpurun_out/r03e_steady_state.err
cat gpurun_out/r03e_gputests.log; head -40 gpurun_out/r03e_steady_state.md
for f in gpurun_out/r03e_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get("roofline",{}).get("by_kind_ms_per_step",{})
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), {x:k.get(x) for x in ("mlp_fused","mlp_bwd_fused","gemm_nt_bf16","attention_bwd","gemm_nt_lnbwd","gemm_tn_atomic","gemm_nt_dgelu")})
except Exception as e:
    print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
