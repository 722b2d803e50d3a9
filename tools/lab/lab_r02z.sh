cd /root/repo; mkdir -p gpurun_out
BENCH_FORCE_DIST=1 MASTER_PORT=29519 timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02z_forcedist.json 2> gpurun_out/r02z_forcedist.err
timeout 300 python bench.py --epoch 30 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02z_epoch30.json 2> /dev/null
timeout 300 python bench.py --arch vit_base --batch 128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02z_vit_base.json 2> /dev/null
timeout 300 python bench.py --workload finetune --batch 512 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02z_finetune.json 2> gpurun_out/r02z_finetune.err
python - <<PY
import json
for f in ["forcedist","epoch30","vit_base","finetune"]:
    try:
        d=json.loads(open("gpurun_out/r02z_%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"))
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/r02z_finetune.err
