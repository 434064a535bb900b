#!/bin/bash
# round 4, GPU call 56: which timed block of the driver-style run is the slow one
set -o pipefail
mkdir -p gpurun_out/r4ah
SSA_BENCH_BLOCKS=gpurun_out/r4ah/blocks.txt python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r4ah/bench.json 2> gpurun_out/r4ah/bench.err; echo "rc $?"
python3 -c "
import numpy as np
for ln in open('gpurun_out/r4ah/blocks.txt'):
    el=np.array([float(x) for x in ln.split()]); print(len(el), 'blocks; median %.3f ms; slow blocks (index, ms):' % (1e3*np.median(el)), [(int(i), round(1e3*float(el[i]),2)) for i in np.argsort(el)[-4:]])"
