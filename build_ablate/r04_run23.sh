#!/bin/bash
set -u
mkdir -p gpurun_out/r4w
T0=$(date +%s); python3 bench.py > gpurun_out/r4w/bench.json 2> gpurun_out/r4w/bench.err; echo "bench rc $? wall $(( $(date +%s) - T0 )) s" | tee -a gpurun_out/r4w/summary.txt
python3 - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r4w/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], d['value_spread'], 'frac', d['roofline']['frac'], d['roofline']['kernel_ms'])
for k, v in d.items():
    if isinstance(v, dict) and 'value' in v:
        print(k, v['value'], v.get('roofline_frac'), {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv})
    elif isinstance(v, dict):
        sub = {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv}
        if sub: print(k, sub)
PY
