cd /root/repo; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --arch vit_base --batch 128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02h_$tag.json 2> gpurun_out/r02h.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r02h_$tag.json").read().strip().splitlines()[-1]); print("$tag", d["ms_per_step"], d["value"], d["roofline"]["by_kind_ms_per_step"])
PY
}
run mlp0 CCD_FUSE_MLP=0
run mlp0_lnbwd0 CCD_FUSE_MLP=0 CCD_FUSE_LNBWD=0
run mlp0_ln0 CCD_FUSE_MLP=0 CCD_FUSE_LN=0
run lnbwd0 CCD_FUSE_LNBWD=0
