mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2z_pytest.log
for pl in trace aer; do
MASTER_ADDR=127.0.0.1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1500 --warmup 200 --no-cpu-baseline --no-legs --payload $pl > gpurun_out/r2z_rccl1_$pl.json 2> gpurun_out/r2z_rccl1_$pl.err; echo "rc $?"
python - <<PY
import json
d=json.load(open("gpurun_out/r2z_rccl1_$pl.json")); print("$pl", d["value"], d["ms_per_step"], d["config"]["allgather"], d["config"]["allgather_bytes_per_rank"], d["config"]["allgather_warmup_probe"])
PY
done
