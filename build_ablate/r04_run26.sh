#!/bin/bash
mkdir -p gpurun_out/r4z
for p in hybrid fg elements; do PROP=$p timeout -k 10 300 python3 build_ablate/layout_episode_ab.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/r4z/layout_episode_ab.txt; done
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4z/pytest_q.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/r4z/pytest_q.log
