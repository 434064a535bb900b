#!/bin/bash
mkdir -p gpurun_out/r4ii
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x -s > gpurun_out/r4ii/pytest_s.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|^\[hybrid\]|^\[oracle\]" gpurun_out/r4ii/pytest_s.log | cut -c1-420 | tail -5
PROP=hybrid timeout -k 10 300 python3 build_ablate/layout_episode_ab.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4ii/layout_episode_ab.txt
for L in 1; do LAYOUT=$L PROP=hybrid STEPS=330 python3 build_ablate/wave_timeline.py 2>&1 | grep -v amdgpu > gpurun_out/r4ii/wave_timeline_layout$L.txt; done
sed -n 1,5p gpurun_out/r4ii/wave_timeline_layout1.txt; grep "kepler stage by branch" gpurun_out/r4ii/wave_timeline_layout1.txt
