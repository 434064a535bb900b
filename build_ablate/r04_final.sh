#!/bin/bash
# round 4: the evidence of the FINAL build in one call -- profiles/collect.sh r04 (kernel-trace stats + PMC passes of the default variant),
# the other configurations (build_ablate/r04_run18.sh), the -s report of the GPU suite, smoke(), the soaks
set -u
R=$(pwd)
bash profiles/collect.sh r04 > gpurun_out/r04_collect.log 2>&1; echo "collect rc $?"
head -3 gpurun_out/r04/kernel_stats.csv | cut -c1-200
bash build_ablate/r04_run18.sh > gpurun_out/r04_more.log 2>&1; echo "more rc $?"
tail -12 gpurun_out/r04_more.log | cut -c1-250
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_more/smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/r04_more/smoke.log | cut -c1-250
SSA_ALLGATHER=peer python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --no-legs --no-cpu-baseline > gpurun_out/r04_more/bench_hybrid_peer1.json 2> gpurun_out/r04_more/bench_peer1.err; echo "peer1 rc $?"
python3 bench.py > gpurun_out/r04_more/bench_hybrid.json 2> gpurun_out/r04_more/bench_hybrid.err; echo "bench rc $?"
EPISODES=300 PROP=hybrid python3 build_ablate/soak.py > gpurun_out/r04_more/soak_hybrid.txt 2>&1; echo "soak hybrid rc $?"; tail -3 gpurun_out/r04_more/soak_hybrid.txt | cut -c1-250
EPISODES=300 PROP=fg python3 build_ablate/soak.py > gpurun_out/r04_more/soak_fg.txt 2>&1; echo "soak fg rc $?"; tail -3 gpurun_out/r04_more/soak_fg.txt | cut -c1-250
