#!/bin/bash
# round 3, batch h: the training step as one HIP graph (pretrain.GraphedTrainingStep) - parity test, then eager vs graphed at
# 64 and 256 images per GPU (same box, back to back)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export BENCH_NOTIMER=0
timeout 600 python -m pytest tests/test_model_gpu.py -q -x -k "graphed or optimizer_host" 2>&1 | tail -15 > gpurun_out/r03h_graph_test.log
cat gpurun_out/r03h_graph_test.log
: > gpurun_out/r03h_graph_ab.jsonl
for B in 64 256; do
  for G in "" "--graph"; do
    timeout 600 python bench.py --batch $B --steps 20 --warmup 4 --no-cpu-baseline $G 2>gpurun_out/r03h_err_${B}${G}.log | tail -1 >> gpurun_out/r03h_graph_ab.jsonl
    tail -3 gpurun_out/r03h_err_${B}${G}.log
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r03h_graph_ab.jsonl"):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:200]); continue
    print(d["config"]["global_batch"], "graph" if "hip_graph" in d else "eager", d["ms_per_step"], d["value"], d.get("hip_graph"))
PY
