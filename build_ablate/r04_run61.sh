#!/bin/bash
# round 4, GPU call 61: the bench lines of the last commit (default flags; fg)
set -o pipefail
mkdir -p gpurun_out/r4al
python3 bench.py > gpurun_out/r4al/bench_hybrid.json 2> gpurun_out/r4al/bench_hybrid.err; echo "bench rc $?"
python3 bench.py --propagator fg --no-legs > gpurun_out/r4al/bench_fg.json 2> gpurun_out/r4al/bench_fg.err; echo "bench fg rc $?"
python3 -c "
import json
for n in ('hybrid','fg'):
    d=json.load(open('gpurun_out/r4al/bench_%s.json' % n)); print(n, d['value'], d['ms_per_step'], d['value_spread'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))
d=json.load(open('gpurun_out/r4al/bench_hybrid.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'value' in v: print(k, v['value'], v.get('value_spread'), v.get('caller_order'))
g=d['gym_api']; print({k:(v['value'], v['value_spread']) for k,v in g.items() if isinstance(v,dict)})"
