#!/bin/bash
# round 4, GPU call 40: the vector env with and without its per-env storage layout, timed over whole episodes
set -o pipefail
mkdir -p gpurun_out/r4ss
python - <<'PY' > gpurun_out/r4ss/vec_layout.txt 2>&1
import sys, json
sys.argv = ['bench.py']
import bench
for lay in (False, True, False, True):
    r = bench.vec_env_rate(20000, obs_device=True, layout=lay)
    print("vec_env_device_obs layout=%s" % lay, r["value"], r["ms_per_vector_step"], r.get("value_spread"), r["repeats"], flush=True)
for lay in (False, True):
    r = bench.vec_env_rate(20000, layout=lay)
    print("vec_env (host 'aer' observations) layout=%s" % lay, r["value"], r["ms_per_vector_step"], r.get("value_spread"), r["repeats"], flush=True)
PY
echo "vec rc $?"; cat gpurun_out/r4ss/vec_layout.txt | grep -v amdgpu.ids
