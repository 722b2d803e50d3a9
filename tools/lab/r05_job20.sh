#!/bin/bash
# E3: do the 6-us gaps follow the LDS request?  rowgemm (150 KiB, no gaps) with 4 KiB more
mkdir -p gpurun_out
CCD_LAB=8 bash tools/prof_bench.sh r05_lds154 > gpurun_out/r05_lds154.out 2>&1
python - <<'PY'
import re, collections
rows=[]
for l in open('/root/repo/gpurun_out/r05_lds154_sequence.md'):
    m=re.match(r"\| (\d+) \| ([\d.]+) \| `(.*?)` \| (\d+) x (\d+) \| ([\d.]+) \| (-?[\d.]+) \|",l)
    if m: rows.append((int(m[1]),float(m[2]),m[3],float(m[6]),float(m[7])))
c=collections.Counter(); n=collections.Counter()
for i,r in enumerate(rows):
    if r[4]>1.0: c[r[2][:44]]+=r[4]; n[r[2][:44]]+=1
print(len(rows), rows[-1][1], [(k, round(v,1), n[k]) for k,v in c.most_common(8)])
PY
