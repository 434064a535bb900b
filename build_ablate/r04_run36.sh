#!/bin/bash
mkdir -p gpurun_out/r4oo
timeout -k 10 1100 python3 -X faulthandler -m pytest tests -m gpu -q -s -x > gpurun_out/r4oo/pytest_s.log 2>&1; echo "pytest -s rc $?"; grep -E "passed|failed" gpurun_out/r4oo/pytest_s.log | tail -1
python3 bench.py > gpurun_out/r4oo/bench.json 2> gpurun_out/r4oo/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r4oo/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], d['value_spread'], 'frac', d['roofline']['frac'], d['roofline']['kernel_ms'])
for k, v in d.items():
    if isinstance(v, dict) and 'value' in v:
        print(k, v['value'], {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv})
    elif isinstance(v, dict):
        sub = {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv}
        if sub: print(k, sub)
PY
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
