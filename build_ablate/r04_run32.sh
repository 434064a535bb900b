#!/bin/bash
mkdir -p gpurun_out/r4jj
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4jj/pytest_q.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r4jj/pytest_q.log
EPISODES=100 PROP=hybrid python3 build_ablate/soak.py 2>&1 | grep -v amdgpu | cut -c1-260
python3 - <<'PY'
import sys
sys.argv = ['bench.py']
import bench, json
for _ in range(2):
    r = bench.closed_loop_rate(20000, 480, 100, propagator='hybrid')
    print('closed_loop', r['value'], r['value_spread'])
PY
