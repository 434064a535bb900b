#!/bin/bash
# round 4, GPU call 10: the bench line after the lazy failure messages, then the round's profile evidence (kernel-trace stats + PMC passes)
set -o pipefail
mkdir -p gpurun_out/r4j
python bench.py > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err; echo "bench rc $?" | tee -a gpurun_out/r4j/summary.txt
python - <<'PY' | tee -a gpurun_out/r4j/summary.txt
import json
d = json.loads([l for l in open('gpurun_out/r4j/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
for k, v in d['config'].items():
    if isinstance(v, dict) and 'value' in v:
        print(k, v['value'], {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv})
PY
bash profiles/collect.sh r04 > gpurun_out/r4j/collect.log 2>&1; echo "collect rc $?" | tee -a gpurun_out/r4j/summary.txt
tail -40 gpurun_out/r4j/collect.log
