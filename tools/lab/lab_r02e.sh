cd /root/repo; mkdir -p gpurun_out
RG_QUICK=1 timeout 300 python tools/rowgemm_lab.py --rows 131072 2>&1 | tee gpurun_out/rowgemm_lab_c.jsonl | grep -v amdgpu.ids
for v in 1 2; do
CCD_ROWGEMM=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02e_bench_rg$v.json 2> gpurun_out/r02e_bench.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r02e_bench_rg$v.json").read().strip().splitlines()[-1]); print("rowgemm=$v", d["ms_per_step"], d["roofline"]["by_kind_ms_per_step"])
PY
done
CCD_FUSE_LNBWD=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02e_bench_nolnbwd.json 2> gpurun_out/r02e_bench_nolnbwd.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r02e_bench_nolnbwd.json").read().strip().splitlines()[-1]); print("unfused", d["ms_per_step"], d["roofline"]["by_kind_ms_per_step"])
PY
