#!/bin/bash
mkdir -p gpurun_out/r4ll
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4ll/pytest_q.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r4ll/pytest_q.log | cut -c1-250
python3 - <<'PY'
import sys
sys.argv = ['bench.py']
import bench
for lay in (False, True):
    r = bench.closed_loop_rate(20000, 480, 100, propagator='hybrid', layout=lay)
    print('closed_loop persistent layout', lay, r['value'], r['value_spread'], r['distinct_objects_selected'], r['failed_filters'])
    r = bench.closed_loop_rate(20000, 200, 100, propagator='hybrid', persistent=False, layout=lay)
    print('closed_loop per-step   layout', lay, r['value'], r['value_spread'], r['distinct_objects_selected'], r['failed_filters'])
PY
