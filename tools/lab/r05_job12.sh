#!/bin/bash
# round 5, job 12: the fused classifier tail of the segmentation head: parity on the GPU, A/B of the step, launch sequence
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "cls_tail or seghead or conv_pieces or small_step or tiny" > gpurun_out/r05_job12_tests.log 2>&1
tail -4 gpurun_out/r05_job12_tests.log
for rep in 1 2; do
  for f in 1 0; do
    CCD_FUSE_CLS_TAIL=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'cls_tail_fused': $f, 'rep': $rep, 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" | tee -a gpurun_out/r05_cls_tail_ab.jsonl
  done
done
bash tools/prof_bench.sh r05_tail > gpurun_out/r05_tail.out 2>&1
grep -n "cls_tail" gpurun_out/r05_tail_sequence.md gpurun_out/r05_tail_steady_state.md | cut -c1-170
