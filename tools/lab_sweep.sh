export CCD_HIP_LIB=/root/repo/gpurun_lab/libccd_lab.so
for shape in "131072 1152 384" "131072 1536 384" "8192 8192 4096"; do
  LAB_MFAST=64 python tools/gemm_lab.py nt $shape 1 2>&1 | tail -11
done
