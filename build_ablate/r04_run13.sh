#!/bin/bash
# round 4, GPU call 13: run_policy with pipelined chunks and the arg-max head; the whole bench line
set -o pipefail
mkdir -p gpurun_out/r4m
timeout -k 10 500 python -m pytest tests/test_env_gpu.py -m gpu -x -q > gpurun_out/r4m/pytest_env.log 2>&1; echo "pytest env rc $?" | tee -a gpurun_out/r4m/summary.txt
tail -5 gpurun_out/r4m/pytest_env.log
python bench.py > gpurun_out/r4m/bench.json 2> gpurun_out/r4m/bench.err; echo "bench rc $?" | tee -a gpurun_out/r4m/summary.txt
python - <<'PY' | tee -a gpurun_out/r4m/summary.txt
import json
d = json.loads([l for l in open('gpurun_out/r4m/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
for k, v in d.items():
    if isinstance(v, dict) and 'value' in v:
        print(k, v['value'], {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv})
    elif isinstance(v, dict):
        sub = {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv}
        if sub: print(k, sub)
PY
