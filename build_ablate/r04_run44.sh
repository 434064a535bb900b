#!/bin/bash
# round 4, GPU call 44: SSA_LAUNCH_FOLD_INSIDE in the grid-stride instance (tiles counted once per wavefront); suite, vector env, 160 000 objects
set -o pipefail
mkdir -p gpurun_out/r4ww
python -m pytest tests -m gpu -q -x > gpurun_out/r4ww/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4ww/pytest.log
bash build_ablate/r04_run40.sh
python3 bench.py --objects 160000 --no-legs --no-cpu-baseline --steps 479 --warmup 0 > gpurun_out/r4ww/bench160k.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/r4ww/bench160k.json')); print('160k', d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['value_spread'])"
DEV=1 python3 build_ablate/vec_env_host_probe.py 2>&1 | grep "per vector"
