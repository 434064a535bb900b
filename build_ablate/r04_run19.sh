#!/bin/bash
# round 4, GPU call 19: the whole GPU suite with -s (the round's parity report), smoke(), the 1-rank sharded bench lines (RCCL / peer stores),
# soaks of the final build
set -u
R=$(pwd); OUT=$R/gpurun_out/r4s; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -s -x > $OUT/pytest_s.log 2>&1; echo "pytest -s rc $?" | tee -a $OUT/summary.txt
grep -E "^\[|passed|failed|rror" $OUT/pytest_s.log | cut -c1-400 | tail -30
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" | tee -a $OUT/summary.txt; tail -2 $OUT/smoke.log
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-legs --no-cpu-baseline > $OUT/bench_hybrid_rccl1.json 2> $OUT/bench_rccl1.err; echo "bench rccl1 rc $?"
SSA_ALLGATHER=peer python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --no-legs --no-cpu-baseline > $OUT/bench_hybrid_peer1.json 2> $OUT/bench_peer1.err; echo "bench peer1 rc $?"
for f in $OUT/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('allgather_api', '')[:40], d['config'].get('sharded_enqueue'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
EPISODES=300 PROP=hybrid python3 build_ablate/soak.py > $OUT/soak_hybrid.txt 2>&1; echo "soak hybrid rc $?" | tee -a $OUT/summary.txt; tail -6 $OUT/soak_hybrid.txt | cut -c1-300
EPISODES=300 PROP=fg python3 build_ablate/soak.py > $OUT/soak_fg.txt 2>&1; echo "soak fg rc $?" | tee -a $OUT/summary.txt; tail -6 $OUT/soak_fg.txt | cut -c1-300
