#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4ap
python -m pytest tests/test_hip_step.py -m gpu -q -x -k "several_envs or storage_layout" > gpurun_out/r4ap/pytest.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r4ap/pytest.log | cut -c1-220
