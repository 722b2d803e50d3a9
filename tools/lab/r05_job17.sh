#!/bin/bash
# batched launches of the segmentation head (one re-layout launch, one fill, one finalize per level): tests + same-box A/B against the previous seghead.py
mkdir -p gpurun_out
rm -f gpurun_out/r05_seghead_launches_ab.jsonl
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "multi_launch or cls_tail or seghead or conv_pieces or small_step or tiny" > gpurun_out/r05_job17_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r05_job17_tests.log | tail -2
cp ccd_amd/seghead.py /tmp/seghead_new.py
for rep in 1 2; do
  for which in new prev; do
    if [ $which = prev ]; then cp tools/lab/_seghead_prev.py ccd_amd/seghead.py   # (the previous commit's file, placed there for this run only); else cp /tmp/seghead_new.py ccd_amd/seghead.py; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'seghead': '$which', 'rep': $rep, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" | tee -a gpurun_out/r05_seghead_launches_ab.jsonl
  done
done
cp /tmp/seghead_new.py ccd_amd/seghead.py
bash tools/prof_bench.sh r05_batched > gpurun_out/r05_batched.out 2>&1
head -3 gpurun_out/r05_batched_sequence.md
