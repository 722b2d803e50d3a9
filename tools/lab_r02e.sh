cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "lnbwd or small3 or full_batch" 2>&1 | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err; tail -c 900 gpurun_out/r02e_bench.json
CCD_FUSE_LNBWD=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02e_bench_nolnbwd.json 2> gpurun_out/r02e_bench_nolnbwd.err; tail -c 900 gpurun_out/r02e_bench_nolnbwd.json
