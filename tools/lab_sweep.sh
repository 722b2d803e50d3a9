for mode in 0 1 2; do
  echo "CCD_GEMM_ROW384=$mode"
  CCD_GEMM_ROW384=$mode python tools/microbench.py 2>&1 | grep "gemm_nt_proj\|gemm_nt_fc2"
done
for mode in 0 1 2; do
  CCD_GEMM_ROW384=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('row384=$mode', d['value'], d['ms_per_step'], d['roofline']['by_kind_ms_per_step'])"
done
