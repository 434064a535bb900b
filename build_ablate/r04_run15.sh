#!/bin/bash
# round 4, GPU call 15: GPU timeline of run_policy after the pipelined chunks, with the arg-max head
set -o pipefail
R=$(pwd)
mkdir -p gpurun_out/r4o
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4o/trace -- python3 $R/build_ablate/run_policy_trace.py > $R/gpurun_out/r4o/trace.log 2>&1; echo "trace rc $?" | tee -a $R/gpurun_out/r4o/summary.txt
cd $R
python3 build_ablate/run_policy_trace.py --reduce gpurun_out/r4o/trace | tee gpurun_out/r4o/run_policy_timeline.txt
find gpurun_out/r4o/trace -name "*.csv" -size +20M -delete
