#!/bin/bash
# round 4, GPU call 55: the bench exactly as the driver runs it (legs included), timed
set -o pipefail
mkdir -p gpurun_out/r4ag
S=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4ag/bench_driver.json 2> gpurun_out/r4ag/bench_driver.err; echo "bench rc $?"
E=$(date +%s.%N); python3 -c "print('wall %.1f s' % ($E - $S))"
python3 -c "
import json; d=json.load(open('gpurun_out/r4ag/bench_driver.json')); print(d['value'], d['ms_per_step'], d['value_spread'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))
for k,v in d.items():
    if isinstance(v,dict) and 'value' in v: print(k, v['value'])"
