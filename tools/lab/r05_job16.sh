#!/bin/bash
# the fused classifier tail in the step: A/B on one box (two repetitions each way), the lab numbers, the launch sequence, the GPU tests
mkdir -p gpurun_out
rm -f gpurun_out/r05_cls_tail_ab.jsonl gpurun_out/r05_cls_tail_lab.jsonl
for rep in 1 2; do
  for f in 1 0; do
    CCD_FUSE_CLS_TAIL=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'cls_tail_fused': $f, 'rep': $rep, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" | tee -a gpurun_out/r05_cls_tail_ab.jsonl
  done
done
timeout 600 python tools/cls_tail_lab.py 2>&1 | tail -1 | tee -a gpurun_out/r05_cls_tail_lab.jsonl
bash tools/prof_bench.sh r05_tail > gpurun_out/r05_tail.out 2>&1
grep -n "cls_tail" gpurun_out/r05_tail_steady_state.md | cut -c1-170
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05_job16_tests.log 2>&1
tail -3 gpurun_out/r05_job16_tests.log
