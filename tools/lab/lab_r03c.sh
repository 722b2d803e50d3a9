#!/bin/bash
# round 3, GPU batch c: qkv-bias gradient (q from the dQ kernel, v analytic), ln_bwd beside the weight-gradient kernel
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention or layernorm or rowproj" 2>&1 | tail -4 > gpurun_out/r03c_kern.log
python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "tiny_training or two_iterations or four_iterations or one_rank or micro_batches" 2>&1 | tail -4 > gpurun_out/r03c_model.log
ATTN_ONLY=1 python tools/rowproj_lab.py > gpurun_out/r03c_attn_lab.jsonl 2> gpurun_out/r03c_attn_lab.err
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r03c_bench_$name.json 2> gpurun_out/r03c_bench_$name.err
}
run default CCD_X=0
run unfused_side CCD_FUSE_LNBWD=0 CCD_SIDE_STREAM=1
run unfused CCD_FUSE_LNBWD=0
run fused_side CCD_SIDE_STREAM=1
run unfused_side_notimer CCD_FUSE_LNBWD=0 CCD_SIDE_STREAM=1 BENCH_NOTIMER=1
cd /tmp && export TMPDIR=/tmp
CCD_FUSE_LNBWD=0 CCD_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03c_side -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timer --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r03c_prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/prof_r03c_side/* | head; du -sh gpurun_out/prof_r03c_side
cat gpurun_out/r03c_kern.log gpurun_out/r03c_model.log gpurun_out/r03c_attn_lab.jsonl
for f in gpurun_out/r03c_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    k=r.get("by_kind_ms_per_step",{})
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], {x:k.get(x) for x in ("mlp_fused","gemm_nt_bf16","attention_bwd","gemm_nt_lnbwd","layernorm_bwd","gemm_tn_atomic","gemm_nt_dgelu")})
except Exception as e:
    print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
done
