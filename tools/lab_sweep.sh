export CCD_HIP_LIB=/root/repo/gpurun_lab/libccd_lab.so
for shape in "131072 1152 384"; do
  LAB_MFAST=64 python tools/gemm_lab.py nt $shape 1 2>&1 | tail -11
done
unset CCD_HIP_LIB
python tools/microbench.py 2>&1 | grep gemm
