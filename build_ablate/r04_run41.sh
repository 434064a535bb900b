#!/bin/bash
# round 4, GPU call 41: where a vector-env step's time goes (wall vs kernels)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4tt
for L in 0 1; do
LAYOUT=$L EPISODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4tt/prof_$L -- python3 $R/build_ablate/vec_env_probe.py > $R/gpurun_out/r4tt/prof_$L.txt 2>&1; echo "prof $L rc $?"; grep episode $R/gpurun_out/r4tt/prof_$L.txt
for f in $R/gpurun_out/r4tt/prof_$L/*/*kernel_stats.csv; do head -8 $f | cut -c1-200; done
done
