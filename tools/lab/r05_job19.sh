#!/bin/bash
# where do the 6-us gaps around the fused block half come from?  E1: the same trace with CCD_FUSE_PROJ=0 (kernels without scratch)
mkdir -p gpurun_out
CCD_FUSE_PROJ=0 bash tools/prof_bench.sh r05_noproj > gpurun_out/r05_noproj.out 2>&1
python - <<'PY'
import re, collections
for tag in ("r05_noproj",):
    rows=[]
    for l in open(f'/root/repo/gpurun_out/{tag}_sequence.md'):
        m=re.match(r"\| (\d+) \| ([\d.]+) \| `(.*?)` \| (\d+) x (\d+) \| ([\d.]+) \| (-?[\d.]+) \|",l)
        if m: rows.append((int(m[1]),float(m[2]),m[3],float(m[6]),float(m[7])))
    c=collections.Counter()
    for r in rows:
        if r[4]>1.0: c[r[2][:44]]+=r[4]
    print(tag, len(rows), rows[-1][1], c.most_common(6))
PY
