#!/bin/bash
# round 5, GPU job 2: forward-split A/B with the pass on a non-blocking work stream
cd "$(dirname "$0")/../.."
O=gpurun_out
python bench.py --no-cpu-baseline 2>$O/r05_base2.err | tail -1 > $O/r05_base2.json
for s in cu:16 xcd:4 xcd:3 cu:12 xcd:5; do
  CCD_FWD_SPLIT=$s timeout 300 python bench.py --no-cpu-baseline 2>$O/r05_split2_$s.err | tail -1 > $O/r05_split2_$s.json
done
python bench.py --no-cpu-baseline 2>>$O/r05_base2.err | tail -1 > $O/r05_base2b.json
for f in $O/r05_base2.json $O/r05_base2b.json $O/r05_split2_*.json; do echo "$f: $(python -c "
import json,sys
try:
    d=json.load(open('$f')); print(d['ms_per_step'], d['config']['final_loss'])
except Exception as e: print('ERR', e)")"; done
tail -n 3 $O/r05_split2_*.err | tail -40
