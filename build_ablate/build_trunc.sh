#!/bin/bash
# truncation builds of the step kernel: build_ablate/trunc/t<k>.so ends every wave at timeline marker k (1 load, 2 chol, 3 kepler+stage,
# 4 moment sums, 5 covariance, 6 update/status, 7 observe+aer, 8 store, 9 statistics)
R=$(cd "$(dirname "$0")/.." && pwd)
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=fast -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=8"
for k in 1 2 3 4 5 6 7 8 9; do
  /opt/rocm/bin/hipcc $F -DSSA_TRUNC=$k -o $R/build_ablate/trunc/t$k.so $R/ssa-gym_amd/csrc/ssa_kernels.hip 2>/dev/null &
  if [ $((k % 4)) = 0 ]; then wait; fi
done
wait
ls -la $R/build_ablate/trunc
