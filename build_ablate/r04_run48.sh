#!/bin/bash
# round 4, GPU call 48: healthy-phase cost of hybrid vs fg by covariance form
set -o pipefail
mkdir -p gpurun_out/r4zz
python3 build_ablate/healthy_phase_ab.py > gpurun_out/r4zz/healthy_phase_ab.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids gpurun_out/r4zz/healthy_phase_ab.txt
