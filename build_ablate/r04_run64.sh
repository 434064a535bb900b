#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4ao
python3 build_ablate/gym_host_probe.py > gpurun_out/r4ao/gym_host.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids gpurun_out/r4ao/gym_host.txt | head -34 | cut -c1-180
DEV=1 python3 build_ablate/gym_host_probe.py 2>&1 | grep "per step"
