#!/bin/bash
mkdir -p gpurun_out/r4nn
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4nn/pytest_q.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r4nn/pytest_q.log | cut -c1-250
python3 - <<'PY'
import sys
sys.argv = ['bench.py']
import bench
for name, fn in (("flatten", lambda: bench.gym_api_rate(20000, 'flatten')), ("flatten_f32", lambda: bench.gym_api_rate(20000, 'flatten', f32=True)),
                 ("aer", lambda: bench.gym_api_rate(20000, 'aer')), ("aer_f32", lambda: bench.gym_api_rate(20000, 'aer', f32=True)),
                 ("vec_env", lambda: bench.vec_env_rate(20000)), ("vec_env_f32", lambda: bench.vec_env_rate(20000, f32=True))):
    r = fn()
    print(name, r['value'], r['value_spread'], r.get('ms_per_step', r.get('ms_per_vector_step')))
PY
