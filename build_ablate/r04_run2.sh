#!/bin/bash
# round 4, GPU call 2: workgroup-pooled out-of-line propagation -- tests, A/B against round 3, timelines; the ladder's non-monotone case
set -o pipefail
mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -x -q > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4b/summary.txt
tail -3 gpurun_out/r4b/pytest.log
for prop in hybrid fg elements; do
  LIB=build_ablate/libs/r03.so PROP=$prop OUT=/tmp/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4b/summary.txt
  LIB=ssa-gym_amd/libssa_hip.so PROP=$prop REF=/tmp/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -2 | tee -a gpurun_out/r4b/summary.txt
done
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=400 python build_ablate/wave_timeline.py > gpurun_out/r4b/wave_timeline_hybrid_step400.txt 2>&1; echo "timeline rc $?" | tee -a gpurun_out/r4b/summary.txt
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=100 python build_ablate/wave_timeline.py > gpurun_out/r4b/wave_timeline_hybrid_step100.txt 2>&1
head -30 gpurun_out/r4b/wave_timeline_hybrid_step400.txt
# the ladder's non-monotone case: two-pass search (diagnostic build) against the shipped ladder
LIB=build_ablate/libs/twopass.so OUT=/tmp/ladder_twopass.npz python build_ablate/ladder_ab.py 2>&1 | tail -2 | tee -a gpurun_out/r4b/summary.txt
LIB=ssa-gym_amd/libssa_hip.so REF=/tmp/ladder_twopass.npz OUT=gpurun_out/r4b/ladder_case.npz python build_ablate/ladder_ab.py > gpurun_out/r4b/ladder_ab.txt 2>&1; tail -12 gpurun_out/r4b/ladder_ab.txt
