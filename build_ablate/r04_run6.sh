#!/bin/bash
# round 4, GPU call 6: priority change; the ladder case: factors of the three ladder forms; first bench line of the round
set -o pipefail
mkdir -p gpurun_out/r4f
python -m pytest tests -m gpu -q -x > gpurun_out/r4f/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4f/summary.txt
grep -E "passed|failed|FAILED" gpurun_out/r4f/pytest.log | tail -5
LIB=ssa-gym_amd/libssa_hip.so PROP=hybrid python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4f/summary.txt
LIB=ssa-gym_amd/libssa_hip.so PROP=elements python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4f/summary.txt
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=400 python build_ablate/wave_timeline.py > gpurun_out/r4f/wave_timeline_hybrid_step400.txt 2>&1; echo "timeline rc $?" | tee -a gpurun_out/r4f/summary.txt
head -32 gpurun_out/r4f/wave_timeline_hybrid_step400.txt
# the ladder case (saved by call 5): factor rows of the one-pass ladder, the four-rungs-per-pass ladder and the two-pass search
CASE=build_ablate/ladder_case_tile.npz
LIB=ssa-gym_amd/libssa_hip.so CASE=$CASE UOUT=/tmp/U_onepass.npy python build_ablate/ladder_probe_tile.py 2>&1 | tail -6 | tee -a gpurun_out/r4f/summary.txt
LIB=build_ablate/libs/bypasses.so CASE=$CASE UREF=/tmp/U_onepass.npy python build_ablate/ladder_probe_tile.py 2>&1 | tail -12 | tee -a gpurun_out/r4f/summary.txt
LIB=build_ablate/libs/twopass.so CASE=$CASE UREF=/tmp/U_onepass.npy python build_ablate/ladder_probe_tile.py 2>&1 | tail -30 | tee -a gpurun_out/r4f/summary.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; echo "bench rc $?" | tee -a gpurun_out/r4f/summary.txt
python -c "
import json; d=json.load(open('gpurun_out/r4f/bench.json'))
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','frac','kernel_ms','flatten','aer','graph_error')}) for k,v in d.items() if k in ('value','ms_per_step','roofline','hybrid','elements','j2','rollout','closed_loop','closed_loop_torch_policy','vec_env','vec_env_zero_copy','vec_env_device_obs','gym_api','cpu_baseline')})
" | tee -a gpurun_out/r4f/summary.txt
