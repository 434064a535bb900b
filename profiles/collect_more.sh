#!/bin/bash
# More evidence of the round's final build (MI355X box):  bash profiles/collect_more.sh r02
#   HBM traffic (two PMC passes each, calibrated as collect.sh) of the step kernel at 160 000 objects and for the J2 / ELEMENTS
#   propagators at 20 000; kernel-trace statistics of the ELEMENTS and J2 bench commands; per-stage instruction counts
#   (healthy and step-380 states); the per-window episode profile; the A/B of the round-1 library against the final one.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}_more
mkdir -p $OUT
export TAG
cd /tmp && export TMPDIR=/tmp
for cfg in "hybrid 160000" "fg 20000" "fg 160000" "j2 20000" "elements 20000"; do
  set -- $cfg; export PROP=$1 M=$2
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${PROP}_${M}_$c -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_${PROP}_${M}_$c.log 2>&1
  done
  (cd $R && python3 profiles/pmc_reduce.py gpurun_out/${TAG}_more/pmc_${PROP}_${M}_FETCH_SIZE gpurun_out/${TAG}_more/pmc_${PROP}_${M}_WRITE_SIZE > $OUT/traffic_${PROP}_${M}.json 2> $OUT/traffic_${PROP}_${M}.err)
  echo "traffic $PROP $M done"
done
unset PROP M
for prop in fg elements j2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$prop -- python3 $R/bench.py --propagator $prop --steps 958 --warmup 0 --no-cpu-baseline --no-legs --rollout 60 > $OUT/prof_$prop.json 2> $OUT/prof_$prop.err
  find $OUT/prof_$prop -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$prop.csv \;
  head -3 $OUT/kernel_stats_$prop.csv
done
cd $R
bash build_ablate/trunc_counters.sh > $OUT/trunc_healthy.log 2>&1; cp gpurun_out/trunc_stages.txt $OUT/stage_counts_healthy.txt; tail -12 $OUT/stage_counts_healthy.txt
python3 build_ablate/episode_profile.py > $OUT/episode_profile.txt 2>&1; tail -12 $OUT/episode_profile.txt
FAST=1 PROPS=fg python3 build_ablate/time_variants.py > $OUT/variants_20k.txt 2>&1; cat $OUT/variants_20k.txt
FAST=1 M=2000 PROPS=fg python3 build_ablate/time_variants.py > $OUT/variants_2k.txt 2>&1; cat $OUT/variants_2k.txt
# round 3: the closed loop's timeline (two consecutive steps of closed_loop_kernel), the host-path probe, episode-level failures
python3 build_ablate/closed_loop_timeline.py > $OUT/closed_loop_timeline.txt 2>&1; tail -30 $OUT/closed_loop_timeline.txt
python3 build_ablate/host_path_probe.py > $OUT/host_path_probe.txt 2>&1; cat $OUT/host_path_probe.txt
python3 build_ablate/episode_failures.py > $OUT/episode_failures.txt 2>&1; tail -12 $OUT/episode_failures.txt
build_ablate/probe/atomic_rtt > $OUT/atomic_rtt.txt 2>&1
cat profiles/traffic.json | tail -30
cp $R/profiles/traffic.json $OUT/traffic_merged.json
