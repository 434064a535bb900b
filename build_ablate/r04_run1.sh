#!/bin/bash
# round 4, GPU call 1: the one-pass ladder -- tests, A/B against the round-3 library, late-episode timeline
set -o pipefail
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4a/summary.txt
for prop in hybrid fg elements; do
  LIB=build_ablate/libs/r03.so PROP=$prop OUT=gpurun_out/r4a/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -2 | tee -a gpurun_out/r4a/summary.txt
  LIB=ssa-gym_amd/libssa_hip.so PROP=$prop REF=gpurun_out/r4a/ab_r03_$prop.npz OUT=gpurun_out/r4a/ab_new_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -3 | tee -a gpurun_out/r4a/summary.txt
done
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=400 python build_ablate/wave_timeline.py > gpurun_out/r4a/wave_timeline_hybrid_step400.txt 2>&1; echo "timeline rc $?" | tee -a gpurun_out/r4a/summary.txt
head -40 gpurun_out/r4a/wave_timeline_hybrid_step400.txt
