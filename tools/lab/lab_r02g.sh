cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "resid_ln or lnbwd or mlp_fused or other_archs or small3" 2>&1 | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02g_bench.json 2> gpurun_out/r02g_bench.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r02g_bench.json").read().strip().splitlines()[-1]); print("small", d["ms_per_step"], d["value"], d["roofline"]["by_kind_ms_per_step"])
PY
CCD_ROWGEMM=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02g_bench_rg0.json 2> gpurun_out/r02g_bench.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r02g_bench_rg0.json").read().strip().splitlines()[-1]); print("small rowgemm=0", d["ms_per_step"], d["value"], d["roofline"]["by_kind_ms_per_step"])
PY
timeout 300 python bench.py --arch vit_base --batch 128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02g_bench_base.json 2> gpurun_out/r02g_bench_base.err; tail -2 gpurun_out/r02g_bench_base.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r02g_bench_base.json").read().strip().splitlines()[-1]); print("base", d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), d["roofline"]["by_kind_ms_per_step"])
PY
CCD_FUSE_LN=0 CCD_FUSE_LNBWD=0 timeout 300 python bench.py --arch vit_base --batch 128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02g_bench_base_unfused.json 2> gpurun_out/r02g_bench_base.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r02g_bench_base_unfused.json").read().strip().splitlines()[-1]); print("base unfused", d["ms_per_step"], d["value"])
PY
