#!/bin/bash
# round 5, GPU job 1: baseline of the round-4 kernels on this box + the CU-mask probe + forward-split A/B
cd "$(dirname "$0")/../.."
O=gpurun_out
python bench.py --no-cpu-baseline 2>$O/r05_base.err | tail -1 > $O/r05_base.json
timeout 300 python tools/cu_mask_probe.py cu:16 > $O/r05_cu_mask_probe.jsonl 2>$O/r05_cu_mask_probe.err
for s in cu:16 cu:15 cu:14; do
  CCD_FWD_SPLIT=$s timeout 300 python bench.py --no-cpu-baseline 2>$O/r05_split_$s.err | tail -1 > $O/r05_split_$s.json
done
# whole-XCD partitions last, each under its own timeout (a queue without CUs on some XCD may never drain)
timeout 120 python tools/cu_mask_probe.py xcd:4 >> $O/r05_cu_mask_probe.jsonl 2>>$O/r05_cu_mask_probe.err
echo "xcd probe rc=$?" >> $O/r05_cu_mask_probe.err
if tail -1 $O/r05_cu_mask_probe.jsonl | grep -q '"xcd:4"'; then
  for s in xcd:4 xcd:3; do
    CCD_FWD_SPLIT=$s timeout 300 python bench.py --no-cpu-baseline 2>$O/r05_split_$s.err | tail -1 > $O/r05_split_$s.json
  done
fi
for f in $O/r05_base.json $O/r05_split_*.json; do echo "$f: $(python -c "
import json,sys
try:
    d=json.load(open('$f')); print(d['ms_per_step'], d['config']['final_loss'])
except Exception as e: print('ERR', e)")"; done
cat $O/r05_cu_mask_probe.jsonl
tail -3 $O/*.err | tail -40
