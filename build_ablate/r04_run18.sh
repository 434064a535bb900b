#!/bin/bash
# round 4, GPU call 18: the rest of the round's evidence -- PMC traffic of the other configurations, kernel-trace stats of the other variants,
# the per-window episode profile, the bench lines at 2 000 / 160 000 objects and through the 1-rank sharded path (RCCL and peer stores), the
# -s report of the GPU tests
set -u
TAG=r04
export TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_more
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "hybrid 160000" "fg 20000" "fg 160000" "j2 20000" "elements 20000"; do
  set -- $cfg; export PROP=$1 M=$2
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${PROP}_${M}_$c -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_${PROP}_${M}_$c.log 2>&1
  done
  (cd $R && python3 profiles/pmc_reduce.py gpurun_out/r04_more/pmc_${PROP}_${M}_FETCH_SIZE gpurun_out/r04_more/pmc_${PROP}_${M}_WRITE_SIZE > $OUT/traffic_${PROP}_${M}.json 2> $OUT/traffic_${PROP}_${M}.err)
  echo "traffic $PROP $M done"
done
unset PROP M
for prop in fg elements j2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$prop -- python3 $R/bench.py --propagator $prop --steps 958 --warmup 0 --no-cpu-baseline --no-legs --rollout 60 > $OUT/prof_$prop.json 2> $OUT/prof_$prop.err
  find $OUT/prof_$prop -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$prop.csv \;
  head -3 $OUT/kernel_stats_$prop.csv | cut -c1-200
done
cd $R
cp profiles/traffic.json $OUT/traffic_merged.json
find $OUT -name "*.csv" -size +3M -delete
PROP=hybrid python3 build_ablate/episode_profile.py > $OUT/episode_profile_hybrid.txt 2>&1; tail -12 $OUT/episode_profile_hybrid.txt
python3 bench.py --objects 2000 --no-legs > $OUT/bench_hybrid_2k.json 2> $OUT/bench_2k.err; echo "bench 2k rc $?"
python3 bench.py --objects 160000 --no-legs > $OUT/bench_hybrid_160k.json 2> $OUT/bench_160k.err; echo "bench 160k rc $?"
python3 bench.py --propagator fg --no-legs > $OUT/bench_fg.json 2> $OUT/bench_fg.err; echo "bench fg rc $?"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-legs --no-cpu-baseline > $OUT/bench_hybrid_rccl1.json 2> $OUT/bench_rccl1.err; echo "bench rccl1 rc $?"
SSA_ALLGATHER=peer python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --no-legs --no-cpu-baseline > $OUT/bench_hybrid_peer1.json 2> $OUT/bench_peer1.err; echo "bench peer1 rc $?"
python3 bench.py --steps 20 --warmup 5 --no-legs > $OUT/bench_hybrid_steps20.json 2> $OUT/bench_steps20.err; echo "bench steps20 rc $?"
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -s -x > $OUT/pytest_s.log 2>&1; echo "pytest -s rc $?"
grep -E "^\[|passed|failed" $OUT/pytest_s.log | cut -c1-300 | tail -40
for f in $OUT/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config'].get('allgather_api'), d['config'].get('sharded_enqueue'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
