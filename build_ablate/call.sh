mkdir -p gpurun_out
SECONDS=0
timeout -k 10 900 python bench.py > gpurun_out/r2z_bench.json 2> gpurun_out/r2z_bench.err; echo "bench rc $? in $SECONDS s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2z_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], {k:((d.get(k) or {}).get("value")) for k in ("rollout","j2","elements","resample","closed_loop","vec_env")}, (d.get("gym_api") or {}).get("flatten",{}).get("value"), (d.get("gym_api") or {}).get("aer",{}).get("value"))
PY
