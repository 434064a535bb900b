#!/bin/bash
# round 4, GPU call 16: the whole GPU suite, the bench line (regime_sorted leg, arg-max head), run_policy timeline with the new arg-max kernel
set -o pipefail
R=$(pwd)
mkdir -p gpurun_out/r4p
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4p/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4p/summary.txt
tail -5 gpurun_out/r4p/pytest.log
python bench.py > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err; echo "bench rc $?" | tee -a gpurun_out/r4p/summary.txt
python - <<'PY' | tee -a gpurun_out/r4p/summary.txt
import json
d = json.loads([l for l in open('gpurun_out/r4p/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
for k, v in d.items():
    if isinstance(v, dict) and 'value' in v:
        print(k, v['value'], {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv})
    elif isinstance(v, dict):
        sub = {kk: vv['value'] for kk, vv in v.items() if isinstance(vv, dict) and 'value' in vv}
        if sub: print(k, sub)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4p/trace -- python3 $R/build_ablate/run_policy_trace.py > $R/gpurun_out/r4p/trace.log 2>&1; echo "trace rc $?" | tee -a $R/gpurun_out/r4p/summary.txt
cd $R
python3 build_ablate/run_policy_trace.py --reduce gpurun_out/r4p/trace > gpurun_out/r4p/run_policy_timeline.txt; grep -A4 "arg-max head" gpurun_out/r4p/run_policy_timeline.txt
find gpurun_out/r4p/trace -name "*.csv" -size +20M -delete
