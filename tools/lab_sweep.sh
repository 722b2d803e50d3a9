for mode in 1 2; do
  echo "CCD_GEMM_256=$mode"
  CCD_GEMM_256=$mode python tools/microbench.py 2>&1 | grep "gemm_nt"
done
