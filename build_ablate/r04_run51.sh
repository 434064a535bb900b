#!/bin/bash
# round 4, GPU call 51: reference covariance -- the mean reaches the y rows by DPP broadcasts (+ -DSSA_COV_LEAN): bit-identical? faster?
set -o pipefail
mkdir -p gpurun_out/r4ad
LIB=build_ablate/libs/covlean2.so PROP=hybrid REF=gpurun_out/r4ab/ref_hybrid.npz python3 build_ablate/ab_episode.py 2>&1 | grep -v amdgpu.ids
LIB=build_ablate/libs/covlean2.so PROP=elements REF=gpurun_out/r4ab/ref_elements.npz python3 build_ablate/ab_episode.py 2>&1 | grep -v amdgpu.ids
LIB=build_ablate/libs/covlean.so PROPS=hybrid COVS=reference python3 build_ablate/healthy_phase_ab.py 2>&1 | grep -v amdgpu.ids
LIB=build_ablate/libs/covlean2.so PROPS=hybrid COVS=reference,centred python3 build_ablate/healthy_phase_ab.py 2>&1 | grep -v amdgpu.ids
LIB=build_ablate/libs/trace.so PROP=hybrid COV=reference STEPS=100 LAYOUT=1 python build_ablate/wave_timeline.py > gpurun_out/r4ad/timeline_reference.txt 2>&1; sed -n 2,16p gpurun_out/r4ad/timeline_reference.txt
