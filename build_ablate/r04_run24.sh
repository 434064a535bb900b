#!/bin/bash
mkdir -p gpurun_out/r4x
for cfg in "one shipped" "one build_ablate/libs/chainweak.so" "two build_ablate/libs/chainweak.so"; do
  set -- $cfg
  if [ "$2" = "shipped" ]; then unset LIB; else export LIB=$2; fi
  MODE=$1 PROP=fg timeout -k 10 180 python3 build_ablate/chain_experiment.py 2>&1 | grep -v amdgpu.ids | tail -2
done
