#!/bin/bash
# round 4, GPU call 7: the ladder's two roundings pinned; hybrid default; failure bookkeeping by device gathers; full bench line
set -o pipefail
mkdir -p gpurun_out/r4g
python -m pytest tests -m gpu -q -s -rA > gpurun_out/r4g/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4g/summary.txt
grep -E "passed|failed|FAILED" gpurun_out/r4g/pytest.log | tail -8
grep -E "^\[ladder\]|^\[hybrid\]|^\[elements\]|^\[oracle\]|^\[catalogue\]" gpurun_out/r4g/pytest.log | tee -a gpurun_out/r4g/summary.txt
LIB=ssa-gym_amd/libssa_hip.so PROP=hybrid python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4g/summary.txt
LIB=ssa-gym_amd/libssa_hip.so PROP=fg python build_ablate/ab_episode.py 2>&1 | tail -1 | tee -a gpurun_out/r4g/summary.txt
LIB=build_ablate/libs/trace.so PROP=hybrid STEPS=400 python build_ablate/wave_timeline.py > gpurun_out/r4g/wave_timeline_hybrid_step400.txt 2>&1; echo "timeline rc $?" | tee -a gpurun_out/r4g/summary.txt
head -32 gpurun_out/r4g/wave_timeline_hybrid_step400.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err; echo "bench rc $?" | tee -a gpurun_out/r4g/summary.txt
tail -5 gpurun_out/r4g/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r4g/bench.json'))
keep=('value','ms_per_step','frac','kernel_ms','flatten','aer','flatten_zero_copy','flatten_device_obs','graph_error','eager','env_side_only','value_spread','failed_filters')
def cut(v): return {kk:(cut(vv) if isinstance(vv,dict) else vv) for kk,vv in v.items() if kk in keep} if isinstance(v,dict) else v
print({k:cut(v) for k,v in d.items() if k in ('value','ms_per_step','value_spread','roofline','fg','hybrid','elements','j2','resample','rollout','closed_loop','closed_loop_per_step_launches','closed_loop_torch_policy','vec_env','vec_env_zero_copy','vec_env_device_obs','gym_api','cpu_baseline','episode_failures')})
" | tee -a gpurun_out/r4g/summary.txt
