cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "lnbwd" 2>&1 | tail -3
RG_QUICK=1 RG_LAB=4 timeout 300 python tools/rowgemm_lab.py --rows 131584 2>&1 | tee gpurun_out/rowgemm_lab_b.jsonl | grep -v amdgpu.ids
RG_PHASES=1 CCD_HIP_LIB=/root/repo/ccd_amd/libccd_lab.so timeout 300 python tools/rowgemm_lab.py --rows 131584 2>&1 | tee gpurun_out/rowgemm_phases_b.jsonl | grep -v amdgpu.ids
