mkdir -p gpurun_out
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2z_smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/r2z_smoke.log
timeout -k 10 900 python bench.py > gpurun_out/r2z_bench.json 2> gpurun_out/r2z_bench.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --objects 2000 --no-cpu-baseline --no-legs > gpurun_out/r2z_2k.json 2> gpurun_out/r2z_2k.err
timeout -k 10 300 python bench.py --objects 64 --no-cpu-baseline --no-legs > gpurun_out/r2z_64.json 2> gpurun_out/r2z_64.err
timeout -k 10 600 python bench.py --objects 160000 --steps 600 --warmup 100 --no-cpu-baseline --no-legs > gpurun_out/r2z_160k.json 2> gpurun_out/r2z_160k.err
MASTER_ADDR=127.0.0.1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1500 --warmup 200 --no-cpu-baseline --no-legs > gpurun_out/r2z_rccl1_trace.json 2> gpurun_out/r2z_rccl1_trace.err
timeout -k 10 600 bash profiles/collect.sh r02 > gpurun_out/r2z_collect.log 2>&1; echo "collect rc $?"
timeout -k 10 900 bash profiles/collect_more.sh r02 > gpurun_out/r2z_more.log 2>&1; echo "more rc $?"
cp ssa-gym_amd/libssa_hip.so /tmp/keep.so; cp build_ablate/libs/trace.so ssa-gym_amd/libssa_hip.so
timeout -k 10 200 python build_ablate/wave_timeline.py > gpurun_out/r2z_tl20k.txt 2>&1
M=64 timeout -k 10 200 python build_ablate/wave_timeline.py > gpurun_out/r2z_tl64.txt 2>&1
STEPS=380 timeout -k 10 200 python build_ablate/wave_timeline.py > gpurun_out/r2z_tl_late.txt 2>&1
cp /tmp/keep.so ssa-gym_amd/libssa_hip.so
python - <<'PY'
import json
for f in ("r2z_bench","r2z_2k","r2z_64","r2z_160k","r2z_rccl1_trace"):
    d=json.load(open("gpurun_out/%s.json"%f))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("fp64_frac"), {k:((d.get(k) or {}).get("value")) for k in ("rollout","j2","elements","resample","closed_loop")}, (d.get("gym_api") or {}).get("flatten",{}).get("value"), (d.get("gym_api") or {}).get("aer",{}).get("value"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline_all_cores") or {}).get("value"))
PY
head -3 gpurun_out/r02/kernel_stats.csv | cut -c1-150
