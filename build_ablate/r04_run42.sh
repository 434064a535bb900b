#!/bin/bash
# round 4, GPU call 42: the vector env's kernel with 'flatten' observations (no az-el-range epilogue), with and without the layout
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4uu
for L in 0 1; do
MODE=flatten LAYOUT=$L EPISODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4uu/prof_$L -- python3 $R/build_ablate/vec_env_probe.py > $R/gpurun_out/r4uu/prof_$L.txt 2>&1; echo "prof $L rc $?"; grep episode $R/gpurun_out/r4uu/prof_$L.txt
for f in $R/gpurun_out/r4uu/prof_$L/*/*kernel_stats.csv; do head -4 $f | cut -c1-200; done
done
python3 $R/bench.py --objects 160000 --no-legs --no-cpu-baseline --steps 479 --warmup 0 > $R/gpurun_out/r4uu/bench160k.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r4uu/bench160k.json')); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['value_spread'])"
