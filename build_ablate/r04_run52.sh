#!/bin/bash
# round 4, GPU call 52: reference covariance, the original form against (six matrix instructions fewer + the mean by DPP broadcasts): bit-identical? faster?
set -o pipefail
mkdir -p gpurun_out/r4ae
for P in hybrid elements; do
LIB=build_ablate/libs/cov_orig.so PROP=$P OUT=gpurun_out/r4ae/ref_$P.npz python3 build_ablate/ab_episode.py 2>&1 | grep -v amdgpu.ids
LIB=build_ablate/libs/cov_new.so PROP=$P REF=gpurun_out/r4ae/ref_$P.npz python3 build_ablate/ab_episode.py 2>&1 | grep -v amdgpu.ids
done
for i in 1 2 3; do
for L in cov_orig cov_new; do
echo -n "$L: "; LIB=build_ablate/libs/$L.so PROPS=hybrid COVS=reference python3 build_ablate/healthy_phase_ab.py 2>&1 | grep -v amdgpu.ids | tail -1
done; done
