#!/bin/bash
# round 4, GPU call 39: the vector env's per-env storage layout (ssa_step_params.obj_ids with several envs)
set -o pipefail
mkdir -p gpurun_out/r4rr
python -m pytest tests/test_env_gpu.py -m gpu -q -x -k "storage_layout or float32 or vector or vec" > gpurun_out/r4rr/pytest_env.log 2>&1; echo "pytest env rc $?"; tail -3 gpurun_out/r4rr/pytest_env.log
python -m pytest tests/test_hip_step.py -m gpu -q -x -k "layout or 160000" > gpurun_out/r4rr/pytest_step.log 2>&1; echo "pytest step rc $?"; tail -3 gpurun_out/r4rr/pytest_step.log
python - <<'PY' > gpurun_out/r4rr/vec_layout.txt 2>&1
import sys, json
sys.argv = ['bench.py']
import bench
for lay in (False, True, False, True):
    r = bench.vec_env_rate(20000, obs_device=True, layout=lay)
    print("vec_env_device_obs layout=%s" % lay, r["value"], r["ms_per_vector_step"], r.get("value_spread"), flush=True)
for lay in (False, True):
    r = bench.vec_env_rate(20000, layout=lay)
    print("vec_env (host 'aer' observations) layout=%s" % lay, r["value"], r["ms_per_vector_step"], r.get("value_spread"), flush=True)
PY
echo "vec rc $?"; cat gpurun_out/r4rr/vec_layout.txt | grep -v amdgpu.ids
