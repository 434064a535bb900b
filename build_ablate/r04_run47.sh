#!/bin/bash
# round 4, GPU call 47: soak of the vector env's per-env storage layout
set -o pipefail
mkdir -p gpurun_out/r4yy
timeout -k 10 500 python3 build_ablate/vec_env_layout_soak.py > gpurun_out/r4yy/vec_env_layout_soak.txt 2>&1; echo "soak rc $?"; grep -v amdgpu.ids gpurun_out/r4yy/vec_env_layout_soak.txt | cut -c1-250
M=10244 E=3 EPISODES=4 STEPS=120 timeout -k 10 300 python3 build_ablate/vec_env_layout_soak.py > gpurun_out/r4yy/vec_env_layout_soak_3x10244.txt 2>&1; echo "soak 3x10244 rc $?"; grep -v amdgpu.ids gpurun_out/r4yy/vec_env_layout_soak_3x10244.txt | cut -c1-250
