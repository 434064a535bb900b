#!/bin/bash
mkdir -p gpurun_out/r4ee
for L in 0 1; do LAYOUT=$L PROP=hybrid STEPS=330 python3 build_ablate/wave_timeline.py 2>&1 | grep -v amdgpu > gpurun_out/r4ee/wave_timeline_hybrid_step330_layout$L.txt; done
head -3 gpurun_out/r4ee/wave_timeline_hybrid_step330_layout1.txt
