#!/bin/bash
# round 3, batch q: gelu(u) stored by the forward kernel (CCD_STORE_GACT=1) against the default, same box, alternating; then the
# other workloads' bench lines on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r03q_store_gact_ab.jsonl
for rep in 1 2; do for G in 0 1; do
  CCD_STORE_GACT=$G timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['by_kind_ms_per_step']
print(json.dumps({'store_gact': $G, 'ms_per_step': d['ms_per_step'], 'mlp_fused': r.get('mlp_fused'), 'gemm_nt_dgelu': r.get('gemm_nt_dgelu'), 'gemm_tn_atomic': r.get('gemm_tn_atomic')}))" | tee -a gpurun_out/r03q_store_gact_ab.jsonl
done; done
timeout 600 python bench.py --batch 64 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03q_bench_b64.json
timeout 600 python bench.py --epoch 30 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03q_bench_epoch30.json
timeout 600 python bench.py --arch vit_base --batch 128 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03q_bench_vit_base_b128.json
timeout 600 python bench.py --arch vit_base_768 --batch 128 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03q_bench_vit_base_768_b128.json
timeout 600 python bench.py --workload finetune --batch 512 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03q_bench_finetune_b512.json
for f in b64 epoch30 vit_base_b128 vit_base_768_b128 finetune_b512; do python -c "
import json; d=json.load(open('gpurun_out/r03q_bench_$f.json')); print('$f', d['ms_per_step'], d['value'], d['config'].get('step_frac_of_mfma_peak'))"; done
