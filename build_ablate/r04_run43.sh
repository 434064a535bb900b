#!/bin/bash
# round 4, GPU call 43: host side of a vector-env step
set -o pipefail
mkdir -p gpurun_out/r4vv
DEV=1 python3 build_ablate/vec_env_host_probe.py > gpurun_out/r4vv/host_dev.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids gpurun_out/r4vv/host_dev.txt | head -40 | cut -c1-200
DEV=0 python3 build_ablate/vec_env_host_probe.py > gpurun_out/r4vv/host_host.txt 2>&1; echo "rc $?"; grep "per vector" gpurun_out/r4vv/host_host.txt
