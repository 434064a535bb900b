#!/bin/bash
# which of the bench's first timed block's 40 ms is it?  (a) as it is, (b) garbage collector off, (c) the first block timed step by step
set -o pipefail
mkdir -p gpurun_out/r4ai; rm -f gpurun_out/r4ai/*.txt
SSA_BENCH_BLOCKS=gpurun_out/r4ai/a.txt python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > /dev/null 2>&1
SSA_BENCH_BLOCKS=gpurun_out/r4ai/b.txt python3 -c "
import gc, sys, runpy
gc.disable()
sys.argv = ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5', '--no-legs', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" > /dev/null 2>&1
SSA_BENCH_BLOCKS=gpurun_out/r4ai/c.txt SSA_BENCH_STEPTIMES=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline 2>&1 >/dev/null | grep "steptimes" | head -3
for f in a b c; do head -1 gpurun_out/r4ai/$f.txt | cut -d' ' -f1-4; done
