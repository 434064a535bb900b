mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r2z_pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2z_smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/r2z_smoke.log
timeout -k 10 900 python bench.py > gpurun_out/r2z_bench.json 2> gpurun_out/r2z_bench.err; echo "bench rc $?"
timeout -k 10 600 bash profiles/collect.sh r02 > gpurun_out/r2z_collect.log 2>&1; echo "collect rc $?"; head -3 gpurun_out/r02/kernel_stats.csv | cut -c1-150
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2z_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"].get("traffic"), {k:((d.get(k) or {}).get("value")) for k in ("rollout","j2","elements","resample","closed_loop")}, (d.get("gym_api") or {}).get("flatten",{}).get("value"), (d.get("gym_api") or {}).get("aer",{}).get("value"))
PY
