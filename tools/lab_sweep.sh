echo "default"; python tools/microbench.py 2>&1 | grep "gemm_nt"
echo "MIN_N=384"; CCD_GEMM_256_MIN_N=384 python tools/microbench.py 2>&1 | grep "gemm_nt_proj\|gemm_nt_fc2"
echo "MIN_N=384 F32"; CCD_GEMM_256_MIN_N=384 CCD_GEMM_256_F32=1 python tools/microbench.py 2>&1 | grep "gemm_nt_proj\|gemm_nt_fc2\|logits"
for cfg in "" "CCD_GEMM_256_MIN_N=384" "CCD_GEMM_256_MIN_N=384 CCD_GEMM_256_F32=1"; do
  env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['by_kind_ms_per_step'])"
done
