#!/bin/bash
# PMC passes over tools/tn384_pmc.py (GPU box): SQ counters, then FETCH_SIZE.  usage: tools/tn384_pmc.sh <tag> [lab]
cd /tmp && export TMPDIR=/tmp
tag=$1; lab=${2:-0}
d=/root/repo/gpurun_out/pmc_${tag}
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d ${d}_sq -o b -- python /root/repo/tools/tn384_pmc.py $lab > ${d}_sq.log 2>&1
python /root/repo/tools/pmc_sq.py ${d}_sq/b_counter_collection.csv | grep -v "at::\|elementwise" > /root/repo/gpurun_out/pmc_${tag}.md
timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d ${d}_lds -o b -- python /root/repo/tools/tn384_pmc.py $lab > ${d}_lds.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${d}_fetch -o b -- python /root/repo/tools/tn384_pmc.py $lab > ${d}_fetch.log 2>&1
python - <<PY >> /root/repo/gpurun_out/pmc_${tag}.md
import csv, collections
csv.field_size_limit(1 << 30)
for sub in ("lds", "fetch"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for r in csv.DictReader(open("${d}_%s/b_counter_collection.csv" % sub)):
            if "tn384" in r["Kernel_Name"] or "gemm_bf16" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e:
        print(sub, "failed", e)
    for k, c in agg.items():
        for n, v in c.items():
            print(f"{k} {n}: per dispatch {sum(v) / len(v):.4g} (n={len(v)})")
PY
cat /root/repo/gpurun_out/pmc_${tag}.md
rm -rf ${d}_sq ${d}_lds ${d}_fetch
