#!/bin/bash
# round 4, GPU call 12: the peer-store all-gather (two ranks on the one card, one rank eager / graphed, bench rehearsal)
set -o pipefail
mkdir -p gpurun_out/r4l
timeout -k 10 500 python -m pytest tests/test_parallel_gloo.py -m gpu -x -q -s > gpurun_out/r4l/pytest_gloo.log 2>&1; echo "pytest gloo rc $?" | tee -a gpurun_out/r4l/summary.txt
tail -15 gpurun_out/r4l/pytest_gloo.log
timeout -k 10 300 python -m pytest tests/test_hip_step.py -m gpu -x -q -k "graphed_sharded or rccl or sharded" > gpurun_out/r4l/pytest_step.log 2>&1; echo "pytest step rc $?" | tee -a gpurun_out/r4l/summary.txt
tail -15 gpurun_out/r4l/pytest_step.log
