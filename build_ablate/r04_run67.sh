#!/bin/bash
# round 4, GPU call 67: EXPERIMENT -- the reference covariance from the first pass's operands (-DSSA_COV_FROM_OPERANDS): timing, the
# behaviour gate and the GPU suite with that library in the product's place (on the box's scratch copy only)
set -o pipefail
mkdir -p gpurun_out/r4ar
for i in 1 2; do
echo -n "product: "; PROPS=hybrid COVS=reference python3 build_ablate/healthy_phase_ab.py 2>&1 | grep -v amdgpu.ids | tail -1
echo -n "operands: "; LIB=build_ablate/libs/cov_ops.so PROPS=hybrid COVS=reference python3 build_ablate/healthy_phase_ab.py 2>&1 | grep -v amdgpu.ids | tail -1
done
cp ssa-gym_amd/libssa_hip.so gpurun_out/r4ar/product.so.keep
cp build_ablate/libs/cov_ops.so ssa-gym_amd/libssa_hip.so
touch ssa-gym_amd/libssa_hip.so
python -m pytest tests/test_episode_failures.py -m gpu -q -s -x > gpurun_out/r4ar/gate.log 2>&1; echo "gate rc $?"; grep -E "^\[|passed|failed|Error|assert" gpurun_out/r4ar/gate.log | cut -c1-400 | tail -30
python -m pytest tests -m gpu -q > gpurun_out/r4ar/suite.log 2>&1; echo "suite rc $?"; tail -15 gpurun_out/r4ar/suite.log | cut -c1-250
rm -f gpurun_out/r4ar/product.so.keep
