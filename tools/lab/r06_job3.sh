#!/bin/bash
# round 6, job 3: the two-workgroups-per-CU model (VERDICT item 1 go / no-go) + the whole GPU suite on the new defaults
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probe/two_wg_probe.hip -o /tmp/two_wg_probe && timeout 300 /tmp/two_wg_probe > gpurun_out/r06_two_wg_probe.jsonl 2> gpurun_out/r06_two_wg_probe.err
cat gpurun_out/r06_two_wg_probe.jsonl; tail -3 gpurun_out/r06_two_wg_probe.err
python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning\|WeightNorm.apply\|^$" | tail -15 > gpurun_out/r06_gputests_mid.log
cat gpurun_out/r06_gputests_mid.log
