for flag in 0 1; do
  echo "CCD_GEMM_256=$flag"
  CCD_GEMM_256=$flag python tools/microbench.py 2>&1 | grep "gemm_nt"
done
