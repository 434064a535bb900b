mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r2z_pytest.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r2z_smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/r2z_smoke.log
timeout -k 10 900 python bench.py > gpurun_out/r2z_bench.json 2> gpurun_out/r2z_bench.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --objects 2000 --no-cpu-baseline --no-legs > gpurun_out/r2z_2k.json 2> gpurun_out/r2z_2k.err; echo "2k rc $?"
timeout -k 10 300 python bench.py --objects 64 --no-cpu-baseline --no-legs > gpurun_out/r2z_64.json 2> gpurun_out/r2z_64.err; echo "64 rc $?"
timeout -k 10 600 python bench.py --objects 160000 --steps 600 --warmup 100 --no-cpu-baseline --no-legs > gpurun_out/r2z_160k.json 2> gpurun_out/r2z_160k.err; echo "160k rc $?"
python - <<'PY'
import json
for f in ("r2z_bench","r2z_2k","r2z_64","r2z_160k"):
    d=json.load(open("gpurun_out/%s.json"%f))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"].get("fp64_frac"), {k:(d[k]["value"] if isinstance(d.get(k),dict) and "value" in d[k] else None) for k in ("rollout","j2","elements","resample","closed_loop")}, d.get("gym_api",{}).get("flatten"), d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_all_cores",{}).get("value"))
PY
