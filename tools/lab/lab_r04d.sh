#!/bin/bash
# round 4, call d: whole GPU suite on the round-4 tree + smoke + the toy's parity budget
mkdir -p gpurun_out/r04d
python -m pytest tests -q -m gpu -x > gpurun_out/r04d/gputests.log 2>&1; tail -5 gpurun_out/r04d/gputests.log
python __graft_entry__.py smoke > gpurun_out/r04d/smoke.log 2>&1; tail -2 gpurun_out/r04d/smoke.log
python tools/parity_budget.py > gpurun_out/r04d/parity_tiny_budget.json 2> gpurun_out/r04d/parity_budget.err; tail -1 gpurun_out/r04d/parity_tiny_budget.json
