#!/bin/bash
# round 4, GPU call 54: the vector env's layout test with its refusals and to_caller_order()
set -o pipefail
mkdir -p gpurun_out/r4af
python -m pytest tests/test_env_gpu.py -m gpu -q -x -k "storage_layout or float32" > gpurun_out/r4af/pytest.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r4af/pytest.log | cut -c1-250
