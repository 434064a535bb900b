#!/bin/bash
# round 4, GPU call 49: the reference-arithmetic covariance with six matrix instructions fewer (-DSSA_COV_LEAN): bit-identical? faster?
set -o pipefail
mkdir -p gpurun_out/r4ab
PROP=hybrid OUT=gpurun_out/r4ab/ref_hybrid.npz python3 build_ablate/ab_episode.py 2>&1 | grep -v amdgpu.ids
LIB=build_ablate/libs/covlean.so PROP=hybrid REF=gpurun_out/r4ab/ref_hybrid.npz python3 build_ablate/ab_episode.py 2>&1 | grep -v amdgpu.ids
PROP=elements OUT=gpurun_out/r4ab/ref_elements.npz python3 build_ablate/ab_episode.py 2>&1 | grep -v amdgpu.ids
LIB=build_ablate/libs/covlean.so PROP=elements REF=gpurun_out/r4ab/ref_elements.npz python3 build_ablate/ab_episode.py 2>&1 | grep -v amdgpu.ids
PROPS=hybrid COVS=reference python3 build_ablate/healthy_phase_ab.py 2>&1 | grep -v amdgpu.ids
LIB=build_ablate/libs/covlean.so PROPS=hybrid COVS=reference python3 build_ablate/healthy_phase_ab.py 2>&1 | grep -v amdgpu.ids
