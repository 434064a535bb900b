#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=fast -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=8"
mkdir -p $R/build_ablate/cl
/opt/rocm/bin/hipcc $F -o $R/build_ablate/cl/a_base.so $R/ssa-gym_amd/csrc/ssa_kernels.hip 2>/dev/null &
/opt/rocm/bin/hipcc $F -DSSA_CL_NOWAIT -o $R/build_ablate/cl/b_nowait.so $R/ssa-gym_amd/csrc/ssa_kernels.hip 2>/dev/null &
/opt/rocm/bin/hipcc $F -DSSA_CL_NOWAIT -DSSA_CL_NOTREE=1 -o $R/build_ablate/cl/c_nowait_notree.so $R/ssa-gym_amd/csrc/ssa_kernels.hip 2>/dev/null &
/opt/rocm/bin/hipcc $F -DSSA_CL_NOWAIT -DSSA_CL_NOTREE=2 -o $R/build_ablate/cl/d_nowait_notree_noscore.so $R/ssa-gym_amd/csrc/ssa_kernels.hip 2>/dev/null &
wait
ls -la $R/build_ablate/cl
