#!/bin/bash
mkdir -p gpurun_out/r4mm
for L in 0 1; do echo "# LAYOUT=$L (the hashes are of the final states in the caller's order: they must not depend on the layout)"; LAYOUT=$L EPISODES=200 PROP=hybrid python3 build_ablate/soak.py 2>&1 | grep -v amdgpu; done > gpurun_out/r4mm/soak_layout.txt
cut -c1-200 gpurun_out/r4mm/soak_layout.txt
python3 bench.py > gpurun_out/r4mm/bench.json 2> gpurun_out/r4mm/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r4mm/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], d['value_spread'], 'frac', d['roofline']['frac'], d['roofline']['kernel_ms'], d['failed_filters'])
for k, v in d.items():
    if isinstance(v, dict) and 'value' in v:
        print(k, v['value'], v.get('roofline_frac'), {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv})
    elif isinstance(v, dict):
        sub = {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv}
        if sub: print(k, sub)
PY
python3 bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r4mm/bench_steps20.json 2>/dev/null; python3 -c "
import json
d = json.loads([l for l in open('gpurun_out/r4mm/bench_steps20.json') if l.startswith('{')][-1])
print('steps20:', d['value'], d['ms_per_step'], d['value_spread'], d['repeats'], d['roofline']['kernel_ms'])"
