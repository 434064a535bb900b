#!/bin/bash
# round 4, GPU call 5: full GPU suite with the committed catalogue; A/B timing; timeline; the ladder case with its whole tile
set -o pipefail
mkdir -p gpurun_out/r4e
python -m pytest tests -m gpu -q > gpurun_out/r4e/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4e/summary.txt
grep -E "passed|failed|FAILED" gpurun_out/r4e/pytest.log | tail -8
for prop in hybrid elements; do
  LIB=build_ablate/libs/r03.so PROP=$prop OUT=/tmp/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4e/summary.txt
  LIB=ssa-gym_amd/libssa_hip.so PROP=$prop REF=/tmp/ab_r03_$prop.npz python build_ablate/ab_episode.py 2>&1 | tail -2 | tee -a gpurun_out/r4e/summary.txt
done
LIB=ssa-gym_amd/libssa_hip.so PROP=fg python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4e/summary.txt
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=400 python build_ablate/wave_timeline.py > gpurun_out/r4e/wave_timeline_hybrid_step400.txt 2>&1; echo "timeline rc $?" | tee -a gpurun_out/r4e/summary.txt
head -32 gpurun_out/r4e/wave_timeline_hybrid_step400.txt
LIB=build_ablate/libs/twopass.so OUT=/tmp/ladder_twopass.npz python build_ablate/ladder_ab.py 2>&1 | tail -1 | tee -a gpurun_out/r4e/summary.txt
LIB=ssa-gym_amd/libssa_hip.so REF=/tmp/ladder_twopass.npz OUT=gpurun_out/r4e/ladder_case.npz python build_ablate/ladder_ab.py > gpurun_out/r4e/ladder_ab.txt 2>&1; grep -E "first difference|ladder probe" gpurun_out/r4e/ladder_ab.txt | tee -a gpurun_out/r4e/summary.txt
LIB=build_ablate/libs/twopass.so CASE=gpurun_out/r4e/ladder_case.npz python build_ablate/ladder_probe_tile.py 2>&1 | tail -5 | tee -a gpurun_out/r4e/summary.txt
LIB=ssa-gym_amd/libssa_hip.so CASE=gpurun_out/r4e/ladder_case.npz python build_ablate/ladder_probe_tile.py 2>&1 | tail -5 | tee -a gpurun_out/r4e/summary.txt
