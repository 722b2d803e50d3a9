for d in 0 1; do
  echo "DEEP=$d"
  CCD_GEMM_256_DEEP=$d python tools/microbench.py 2>&1 | grep "gemm_nt_qkv\|fc1_gelu\|8192x8192\|logits"
done
for d in 0 1 0 1; do
  CCD_GEMM_256_DEEP=$d python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deep=$d', d['value'], d['ms_per_step'], d['roofline']['by_kind_ms_per_step'])"
done
