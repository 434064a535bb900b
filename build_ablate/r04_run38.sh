#!/bin/bash
# round 4, GPU call 38: how the dispatcher spread a late-episode launch over the SIMDs (wave_timeline.py SIMD_MAP=1), with the storage layout
set -o pipefail
mkdir -p gpurun_out/r4qq
for S in 330 400; do
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=$S LAYOUT=1 SIMD_MAP=1 SIMD_MAP_OUT=gpurun_out/r4qq/simd_map_$S.npy python build_ablate/wave_timeline.py > gpurun_out/r4qq/timeline_$S.txt 2>&1; echo "timeline $S rc $?"
done
grep -A30 "XCD of tile 0" gpurun_out/r4qq/timeline_330.txt | cut -c1-300
