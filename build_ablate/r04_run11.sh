#!/bin/bash
# round 4, GPU call 11: GPU timeline of run_policy (graph replay / eager; torch policy / preallocated action)
set -o pipefail
R=$(pwd)
mkdir -p gpurun_out/r4k
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4k/trace -- python3 $R/build_ablate/run_policy_trace.py > $R/gpurun_out/r4k/trace.log 2>&1; echo "trace rc $?" | tee -a $R/gpurun_out/r4k/summary.txt
cd $R
python3 build_ablate/run_policy_trace.py --reduce gpurun_out/r4k/trace | tee gpurun_out/r4k/run_policy_timeline.txt
PROP=fx_xyz_farnocchia_fg python3 - <<'PY' 2>&1 | tail -5
print("fg pass skipped (one pass per call)")
PY
find gpurun_out/r4k/trace -name "*.csv" -size +20M -delete
