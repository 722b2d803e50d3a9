#!/bin/bash
# Same-box A/B of the bench step: the variants run in turn, ROUNDS times (boxes differ by +- 1.5 %, the order on one box does not).
#   usage: tools/lab/ab_bench.sh TAG ROUNDS "NAME[:ENV=VAL,ENV=VAL...]" ... [-- bench.py arguments]
#   e.g.   tools/lab/ab_bench.sh r06_mlp_bwd 2 "two:CCD_FUSE_MLP_BWD=0" "fused:CCD_FUSE_MLP_BWD=1"
#          tools/lab/ab_bench.sh r06_pad 3 "new" "prev:CCD_HIP_LIB=$PWD/ccd_amd/lab_prev.so" -- --arch vit_base_768 --batch 128
# Writes gpurun_out/TAG_NAME_<round>.json (+ .err) and prints ms per step, images/s and the by-kind table's largest entries.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
tag=$1; rounds=$2; shift 2
variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq 1 $rounds); do
  for v in "${variants[@]}"; do
    name=${v%%:*}; envs=""
    [ "$v" != "$name" ] && envs=${v#*:}
    out=gpurun_out/${tag}_${name}_$r
    env $(echo "$envs" | tr ',' ' ') python bench.py --no-cpu-baseline "$@" 2> $out.err | tail -1 > $out.json
    python - $out.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); k = d["roofline"]["by_kind_ms_per_step"]
    top = sorted(k, key=lambda n: -k[n])[:6]
    print(sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], {n: k[n] for n in top})
except Exception as e:
    print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
  done
done
