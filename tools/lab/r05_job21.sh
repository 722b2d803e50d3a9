#!/bin/bash
# do the 6-us gaps around the fused block half survive a HIP-graph replay of the step?
mkdir -p gpurun_out
for m in "" "--graph" "" "--graph"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'mode': '$m' or 'eager', 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" | tee -a gpurun_out/r05_graph_ab.jsonl
done
bash tools/prof_bench.sh r05_graph --graph > gpurun_out/r05_graph.out 2>&1
python - <<'PY'
import re, collections
rows=[]
for l in open('/root/repo/gpurun_out/r05_graph_sequence.md'):
    m=re.match(r"\| (\d+) \| ([\d.]+) \| `(.*?)` \| (\d+) x (\d+) \| ([\d.]+) \| (-?[\d.]+) \|",l)
    if m: rows.append((int(m[1]),float(m[2]),m[3],float(m[6]),float(m[7])))
c=collections.Counter(); n=collections.Counter()
for i,r in enumerate(rows):
    if r[4]>1.0: c[r[2][:44]]+=r[4]; n[r[2][:44]]+=1
print(len(rows), rows[-1][1], [(k, round(v,1), n[k]) for k,v in c.most_common(8)])
PY
head -4 gpurun_out/r05_graph_steady_state.md
