#!/bin/bash
mkdir -p gpurun_out/r4bb
timeout -k 10 600 python3 -m pytest tests/test_parallel_gloo.py -m gpu -x -q > gpurun_out/r4bb/pytest_gloo.log 2>&1; echo "pytest gloo rc $?"; tail -3 gpurun_out/r4bb/pytest_gloo.log
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-legs --no-cpu-baseline > gpurun_out/r4bb/bench_hybrid_rccl1.json 2> gpurun_out/r4bb/bench_rccl1.err; echo "rccl1 rc $?"
SSA_ALLGATHER=peer python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --no-legs --no-cpu-baseline > gpurun_out/r4bb/bench_hybrid_peer1.json 2> gpurun_out/r4bb/bench_peer1.err; echo "peer1 rc $?"
for f in gpurun_out/r4bb/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('storage_layout', '')[:30], d['config'].get('sharded_enqueue'))
PY
done
