#!/bin/bash
# round 4, GPU call 3: inline strong-hyperbolic tier + fast bands -- tests, timing against round 3, timeline
set -o pipefail
mkdir -p gpurun_out/r4c
python -m pytest tests -m gpu -x -q > gpurun_out/r4c/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4c/summary.txt
tail -3 gpurun_out/r4c/pytest.log
for prop in hybrid elements fg; do
  LIB=build_ablate/libs/r03.so PROP=$prop OUT=/tmp/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4c/summary.txt
  LIB=ssa-gym_amd/libssa_hip.so PROP=$prop REF=/tmp/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -2 | tee -a gpurun_out/r4c/summary.txt
done
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=400 python build_ablate/wave_timeline.py > gpurun_out/r4c/wave_timeline_hybrid_step400.txt 2>&1; echo "timeline rc $?" | tee -a gpurun_out/r4c/summary.txt
head -34 gpurun_out/r4c/wave_timeline_hybrid_step400.txt
