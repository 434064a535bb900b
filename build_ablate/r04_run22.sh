#!/bin/bash
set -u
mkdir -p gpurun_out/r4v
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4v/pytest_q.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4v/summary.txt
tail -6 gpurun_out/r4v/pytest_q.log
python3 bench.py --steps 20 --warmup 5 --no-legs > gpurun_out/r4v/bench_steps20.json 2> gpurun_out/r4v/bench_steps20.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r4v/bench_steps20.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['value_spread'], d['repeats'], d['roofline']['kernel_ms'])
PY
