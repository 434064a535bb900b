#!/bin/bash
# round 4, GPU call 60: rocprofv3 kernel-trace statistics of the 160 000-object bench command (the grid-stride instance)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4ak
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4ak/prof -- python3 $R/bench.py --objects 160000 --no-legs --no-cpu-baseline --steps 479 --warmup 0 > $R/gpurun_out/r4ak/bench.json 2> $R/gpurun_out/r4ak/bench.err; echo "rc $?"
for f in $R/gpurun_out/r4ak/prof/*/*kernel_stats.csv; do cp $f $R/gpurun_out/r4ak/kernel_stats_hybrid_160k.csv; head -4 $f | cut -c1-220; done
