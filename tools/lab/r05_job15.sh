#!/bin/bash
# counters of the fused classifier tail on the lab tool: SQ wait split, then HBM fetch / write (separate --pmc passes)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=/root/repo
d=$R/gpurun_out/pmc_tail_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $d -o b -- python $R/tools/cls_tail_lab.py --fused-only --reps 3 > $d.log 2>&1
d2=$R/gpurun_out/pmc_tail_wait
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $d2 -o b -- python $R/tools/cls_tail_lab.py --fused-only --reps 3 > $d2.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  d3=$R/gpurun_out/pmc_tail_$c
  rocprofv3 --pmc $c --output-format csv -d $d3 -o b -- python $R/tools/cls_tail_lab.py --fused-only --reps 3 > $d3.log 2>&1
done
python - <<'PY'
import csv, collections, glob
csv.field_size_limit(1<<30)
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_tail_*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'cls_tail' in r['Kernel_Name']:
            per[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in per.items():
        print(f.split('/')[-3] if 'pmc_tail' in f.split('/')[-3] else f.split('/')[-2], k, {n: round(sum(v)/len(v)) for n, v in c.items()})
PY
