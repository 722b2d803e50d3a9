for lib in "" /root/repo/gpurun_lab/libccd_lab.so; do
  echo "lib=$lib"
  CCD_HIP_LIB=$lib python tools/microbench.py 2>&1 | grep "gemm_nt_qkv\|fc1_gelu\|8192x8192"
  CCD_HIP_LIB=$lib python tools/microbench.py 2>&1 | grep "gemm_nt_qkv\|fc1_gelu\|8192x8192"
done
