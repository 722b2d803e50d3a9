#!/bin/bash
# round 5: the evidence run - whole GPU suite, smoke, PMC traffic + SQ + LDS passes, kernel trace, the bench lines kept under profiles/
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -8 > gpurun_out/r05_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1
bash tools/pmc_traffic.sh r05 > gpurun_out/r05_pmc_traffic.log 2>&1
bash tools/pmc_sq.sh r05 > gpurun_out/r05_pmc_sq.log 2>&1
bash tools/pmc_lds.sh r05 > gpurun_out/r05_pmc_lds.log 2>&1
bash tools/prof_bench.sh r05_bench_b256 > gpurun_out/r05_prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/pmc_traffic_r05.json profiles/pmc_traffic.json          # so that the bench line below carries traffic / step_bytes
python bench.py > gpurun_out/r05_bench_b256.json 2> gpurun_out/r05_bench_b256.err
BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > gpurun_out/r05_bench_forcedist.json 2> gpurun_out/r05_bench_forcedist.err
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
ENVV=(X=1)
run r05_bench_epoch30 --epoch 30
run r05_bench_b64 --batch 64
run r05_bench_vit_base_b128 --arch vit_base --batch 128
run r05_bench_vit_base_768_b128 --arch vit_base_768 --batch 128
run r05_bench_finetune_b512 --workload finetune --batch 512
run r05_bench_b256_again
ENVV=(CCD_FUSE_PROJ=0)
run r05_bench_b256_unfused_proj
ENVV=(X=1)
run r05_bench_b256_again2
ENVV=(CCD_FUSE_PROJ=0)
run r05_bench_b256_unfused_proj2
ENVV=(CCD_FUSE_CLS_TAIL=0)
run r05_bench_b256_unfused_cls_tail
ENVV=(X=1)
run r05_bench_b256_again3
ENVV=(CCD_FUSE_CLS_TAIL=0)
run r05_bench_b256_unfused_cls_tail2
python tools/cls_tail_lab.py 2> /dev/null | tail -1 > gpurun_out/r05_cls_tail_lab.jsonl
python tools/viewmaker_bench.py 2> /dev/null > gpurun_out/r05_viewmaker.jsonl
cat gpurun_out/r05_cls_tail_lab.jsonl gpurun_out/r05_viewmaker.jsonl
cat gpurun_out/r05_gputests.log; tail -3 gpurun_out/r05_smoke.log; tail -5 gpurun_out/r05_pmc_traffic.log | cut -c1-400
head -30 gpurun_out/r05_bench_b256_steady_state.md
for f in gpurun_out/r05_bench_b256.json gpurun_out/r05_bench_forcedist.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), {k:r.get(k) for k in ("kind","kernel","bound","achieved","frac","traffic","step_bytes","avg_launch_ms")}, d.get("cpu_baseline"))
except Exception as e:
    print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
