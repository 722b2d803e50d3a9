#!/bin/bash
# round 5, GPU job 5: the whole GPU suite on the current tree + smoke + bench
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/r05_j5_gputests.log
tail -6 $O/r05_j5_gputests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > $O/r05_j5_smoke.log; cat $O/r05_j5_smoke.log
python bench.py --no-cpu-baseline 2>$O/r05_j5_bench.err | tail -1 > $O/r05_j5_bench.json
python -c "
import json
d=json.load(open('$O/r05_j5_bench.json')); print(d['ms_per_step'], d['config']['final_loss'], d['roofline']['by_kind_ms_per_step'])"
