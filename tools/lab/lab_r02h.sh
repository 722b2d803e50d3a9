cd /root/repo; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --arch vit_base --batch 128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02h_$tag.json 2> gpurun_out/r02h.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r02h_$tag.json").read().strip().splitlines()[-1]); print("$tag", d["ms_per_step"], d["value"], d["config"].get("step_frac_of_mfma_peak"), d["roofline"]["by_kind_ms_per_step"])
PY
}
run default A=1
run g256f32 CCD_GEMM_256_F32=1
run g256f32_row CCD_GEMM_256_F32=1 CCD_GEMM_256_MIN_N=256
