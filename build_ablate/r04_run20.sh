#!/bin/bash
# round 4, GPU call 20: which ingredient makes the failed-capture graph's destructor abort (host-side torch check; the GPU is not involved)
mkdir -p gpurun_out/r4t
for v in "0 -q" "0 -s" "1 -q" "1 -s"; do
  set -- $v
  timeout -k 10 300 python3 -X faulthandler -m pytest tests/test_env_gpu.py -m gpu -x $2 -k "replayed_from_a_graph" > gpurun_out/r4t/guard$1$2.log 2>&1
  echo "no_guard=$1 flags=$2 rc $?" | tee -a gpurun_out/r4t/summary.txt
done
grep -n "what()" gpurun_out/r4t/*.log | cut -c1-200
