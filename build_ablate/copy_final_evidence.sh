#!/bin/bash
# copies what build_ablate/r04_final.sh left under gpurun_out/ into profiles/ (run in the container, after the GPU call)
G=gpurun_out
cp $G/r04/kernel_stats.csv profiles/r04_kernel_stats_hybrid_20k.csv
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r04/counters.json'))
json.dump(d, open('profiles/r04_counters_hybrid.json', 'w'), indent=1)
PY
for v in fg j2 elements; do cp $G/r04_more/kernel_stats_$v.csv profiles/r04_kernel_stats_${v}_20k.csv; done
cp $G/r04_more/bench_fg.json profiles/r04_bench_fg.json
cp $G/r04_more/bench_hybrid.json profiles/r04_bench_hybrid.json
cp $G/r04_more/bench_hybrid_160k.json profiles/r04_bench_hybrid_160k.json
cp $G/r04_more/bench_hybrid_2k.json profiles/r04_bench_hybrid_2k.json
cp $G/r04_more/bench_hybrid_peer1.json profiles/r04_bench_hybrid_peer1.json
cp $G/r04_more/bench_hybrid_rccl1.json profiles/r04_bench_hybrid_rccl1.json
cp $G/r04_more/bench_hybrid_steps20.json profiles/r04_bench_hybrid_steps20_as_the_driver_runs_it.json
cp $G/r04_more/episode_profile_hybrid.txt profiles/r04_episode_profile_hybrid.txt
cp $G/r04_more/soak_hybrid.txt profiles/r04_soak_hybrid.txt
cp $G/r04_more/soak_fg.txt profiles/r04_soak_fg.txt
cp $G/r04_more/traffic_merged.json profiles/traffic.json
