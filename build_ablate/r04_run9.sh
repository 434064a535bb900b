#!/bin/bash
# round 4, GPU call 9: where the host time of env.step() / run_policy goes with the behaviour-faithful default (late-episode steps)
set -o pipefail
mkdir -p gpurun_out/r4i
python -m pytest tests/test_env_gpu.py -m gpu -q -x > gpurun_out/r4i/pytest_env.log 2>&1; echo "pytest env rc $?" | tee -a gpurun_out/r4i/summary.txt
tail -3 gpurun_out/r4i/pytest_env.log
START=20 N=200 python build_ablate/gym_profile.py > gpurun_out/r4i/gym_profile_early.txt 2>&1
START=300 N=170 python build_ablate/gym_profile.py > gpurun_out/r4i/gym_profile_late.txt 2>&1
grep -E "====|tottime|^\s+[0-9]+\s" gpurun_out/r4i/gym_profile_late.txt | head -40
grep -E "^\{" gpurun_out/r4i/gym_profile_late.txt
python build_ablate/run_policy_profile.py > gpurun_out/r4i/run_policy_profile.txt 2>&1; head -40 gpurun_out/r4i/run_policy_profile.txt
