#!/bin/bash
# round 6: the evidence run on the final sources - whole GPU suite, smoke, PMC traffic + SQ + LDS passes, kernel trace + launch sequence,
# the bench lines kept under profiles/, the N > 1 path on one rank, same-box A/Bs of this round's switches
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -8 > gpurun_out/r06_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
bash tools/pmc_traffic.sh r06 > gpurun_out/r06_pmc_traffic.log 2>&1
bash tools/pmc_sq.sh r06 > gpurun_out/r06_pmc_sq.log 2>&1
bash tools/pmc_lds.sh r06 > gpurun_out/r06_pmc_lds.log 2>&1
bash tools/prof_bench.sh r06_bench_b256 > gpurun_out/r06_prof_bench.log 2>&1
BENCH_FORCE_DIST=1 bash tools/prof_bench.sh r06_forcedist > gpurun_out/r06_prof_forcedist.log 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/pmc_traffic_r06.json profiles/pmc_traffic.json          # so that the bench line below carries traffic / step_bytes
python bench.py > gpurun_out/r06_bench_b256.json 2> gpurun_out/r06_bench_b256.err
BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_forcedist.json 2> gpurun_out/r06_bench_forcedist.err
run() { name=$1; shift; env "${ENVV[@]}" python bench.py --no-cpu-baseline "$@" 2> gpurun_out/$name.err | tail -1 > gpurun_out/$name.json; python - gpurun_out/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
ENVV=(X=1)
run r06_bench_epoch30 --epoch 30
run r06_bench_b64 --batch 64
run r06_bench_vit_base_b128 --arch vit_base --batch 128
run r06_bench_vit_base_768_b128 --arch vit_base_768 --batch 128
run r06_bench_finetune_b512 --workload finetune --batch 512
run r06_bench_b256_again
ENVV=(CCD_FUSE_HEAD_LOSS=0); run r06_bench_b256_unfused_head_loss
ENVV=(CCD_G_BF16=0); run r06_bench_b256_g_fp32
ENVV=(CCD_FOLD_TAP=0); run r06_bench_b256_separate_taps
ENVV=(X=1); run r06_bench_b256_again2
ENVV=(CCD_FUSE_HEAD_LOSS=0 CCD_G_BF16=0 CCD_FOLD_TAP=0); run r06_bench_b256_round5_paths
cat gpurun_out/r06_gputests.log; tail -3 gpurun_out/r06_smoke.log; tail -5 gpurun_out/r06_pmc_traffic.log | cut -c1-400
head -30 gpurun_out/r06_bench_b256_steady_state.md
for f in gpurun_out/r06_bench_b256.json gpurun_out/r06_bench_forcedist.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), {k:r.get(k) for k in ("kind","kernel","bound","achieved","frac","traffic","step_bytes","avg_launch_ms")}, d.get("cpu_baseline"))
except Exception as e:
    print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
