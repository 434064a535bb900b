#!/bin/bash
mkdir -p gpurun_out/r4y
timeout -k 10 600 python3 -m pytest tests/test_hip_step.py -m gpu -x -q -k "storage_layout or position or argmax_sigma" > gpurun_out/r4y/pytest.log 2>&1; echo "pytest rc $?"
tail -25 gpurun_out/r4y/pytest.log | cut -c1-300
