#!/bin/bash
# round 4, GPU call 63: the multi-rank launch form rehearsed on the one card (SSA_BENCH_REHEARSAL=1: 2 ranks share cuda:0), RCCL and peer stores
set -o pipefail
mkdir -p gpurun_out/r4an
export HSA_ENABLE_IPC_MODE_LEGACY=0 SSA_BENCH_REHEARSAL=1
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4an/bench_2ranks.json 2> gpurun_out/r4an/bench_2ranks.err; echo "2 ranks rc $?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r4an/bench_2ranks.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['ms_per_step'], d['scaling'], str(d['config'].get('allgather_api',''))[:80], d['config'].get('parallelism'))"
SSA_ALLGATHER=peer timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29672 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4an/bench_2ranks_peer.json 2> gpurun_out/r4an/bench_2ranks_peer.err; echo "2 ranks peer rc $?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r4an/bench_2ranks_peer.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['ms_per_step'], str(d['config'].get('allgather_api',''))[:80])"
