export CCD_HIP_LIB=/root/repo/gpurun_lab/libccd_lab.so
for shape in "131072 1152 384" "131072 384 1536"; do
for mf in 0 2 4 8 16 32 48 56 6 62; do
  LAB_MFAST=$mf python tools/gemm_lab.py nt $shape 20 2>&1 | tail -1 | sed "s/^/lab=$mf /"
done; done
