#!/bin/bash
set -u
mkdir -p gpurun_out/r4cc
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4cc/pytest_q.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r4cc/pytest_q.log
python3 bench.py > gpurun_out/r4cc/bench.json 2> gpurun_out/r4cc/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r4cc/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], d['value_spread'], 'frac', d['roofline']['frac'], d['roofline']['kernel_ms'], d['failed_filters'])
for k, v in d.items():
    if isinstance(v, dict) and 'value' in v:
        print(k, v['value'], v.get('roofline_frac'), {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv})
    elif isinstance(v, dict):
        sub = {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv}
        if sub: print(k, sub)
PY
python3 bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r4cc/bench_steps20.json 2>/dev/null; python3 -c "
import json
d = json.loads([l for l in open('gpurun_out/r4cc/bench_steps20.json') if l.startswith('{')][-1])
print('steps20:', d['value'], d['ms_per_step'], d['value_spread'], d['repeats'], d['roofline']['kernel_ms'])"
