#!/bin/bash
# round 3, batch p: grid-stride BatchNorm + ReLU kernels, column-reduction geometry - tests, kernel trace of the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "seghead or conv or colsum or region or layernorm or loss or optimizer or center" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r03p
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03p -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/r03p_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
tail -c 330 gpurun_out/r03p_bench.json
