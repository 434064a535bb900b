#!/bin/bash
# round 4, GPU call 62: a fold-inside launch carries the previous deferred fold: tests, driver-style bench
set -o pipefail
mkdir -p gpurun_out/r4am
python -m pytest tests/test_hip_step.py -m gpu -q -x -k "deferred or argmax or fold or one_launch" > gpurun_out/r4am/pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r4am/pytest.log | cut -c1-200
for i in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r4am/bench_$i.json 2> gpurun_out/r4am/bench_$i.err; echo "bench rc $?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4am/bench_$i.json')); print(d['value'], d['ms_per_step'], d['value_spread'], d['failed_filters'] if 'failed_filters' in d else '')"
done
python3 build_ablate/first_block_probe.py 2>&1 | grep "20-step"
