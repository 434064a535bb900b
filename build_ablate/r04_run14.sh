#!/bin/bash
# round 4, GPU call 14: health-sorted tiles again, now that the ladder is row-parallel (round 3 rejected them with the sequential ladder)
set -o pipefail
mkdir -p gpurun_out/r4n
for s in 0 1 2; do
  SORT=$s PROP=hybrid python build_ablate/sorted_tiles_experiment.py > gpurun_out/r4n/sorted_$s.txt 2>&1; echo "sort $s rc $?" | tee -a gpurun_out/r4n/summary.txt
  tail -12 gpurun_out/r4n/sorted_$s.txt
done
