#!/bin/bash
# round 4, GPU call 4: lean conic tier (orientation from the state's own vectors) + fast bands + arg-max slots + closed-loop options
set -o pipefail
mkdir -p gpurun_out/r4d
python -m pytest tests -m gpu -q > gpurun_out/r4d/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4d/summary.txt
grep -E "passed|failed|FAILED|Error" gpurun_out/r4d/pytest.log | tail -15
for prop in hybrid elements; do
  LIB=build_ablate/libs/r03.so PROP=$prop OUT=/tmp/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4d/summary.txt
  LIB=ssa-gym_amd/libssa_hip.so PROP=$prop REF=/tmp/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -2 | tee -a gpurun_out/r4d/summary.txt
done
LIB=ssa-gym_amd/libssa_hip.so PROP=fg python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4d/summary.txt
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=400 python build_ablate/wave_timeline.py > gpurun_out/r4d/wave_timeline_hybrid_step400.txt 2>&1; echo "timeline rc $?" | tee -a gpurun_out/r4d/summary.txt
head -34 gpurun_out/r4d/wave_timeline_hybrid_step400.txt
