#!/bin/bash
# round 4, GPU call 45: after the revert of the grid-stride fold: suite, vector env legs, host probe
set -o pipefail
mkdir -p gpurun_out/r4xx
python -m pytest tests -m gpu -q -x > gpurun_out/r4xx/pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r4xx/pytest.log
bash build_ablate/r04_run40.sh
DEV=1 python3 build_ablate/vec_env_host_probe.py 2>&1 | grep "per vector"
DEV=1 LAYOUT=1 python3 build_ablate/vec_env_host_probe.py 2>&1 | grep "per vector"
