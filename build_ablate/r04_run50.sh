#!/bin/bash
# round 4, GPU call 50: per-stage wave timeline, healthy step, centred vs reference covariance
set -o pipefail
mkdir -p gpurun_out/r4ac
for C in centred reference; do
LIB=build_ablate/libs/trace.so PROP=hybrid COV=$C STEPS=100 LAYOUT=1 python build_ablate/wave_timeline.py > gpurun_out/r4ac/timeline_$C.txt 2>&1; echo "$C rc $?"
sed -n 2,16p gpurun_out/r4ac/timeline_$C.txt
done
