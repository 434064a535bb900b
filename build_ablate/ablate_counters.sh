#!/bin/bash
# dynamic per-stage instruction counts of the step kernel: SQ_INSTS_* of builds with one stage compiled out (-DSSA_ABLATE=<bit>:
# 1 Kepler, 2 Cholesky, 4 covariance finalisation, 8 MFMA moment sums, 16 observe, 32 store + statistics), against the full build
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ablate
mkdir -p $OUT
cp $R/ssa-gym_amd/libssa_hip.so /tmp/keep_full.so
cd /tmp && export TMPDIR=/tmp
for v in full a1 a2 a4 a8 a16 a32; do
  if [ $v = full ]; then cp /tmp/keep_full.so $R/ssa-gym_amd/libssa_hip.so; else cp $R/build_ablate/abl/$v.so $R/ssa-gym_amd/libssa_hip.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/$v -- python3 $R/profiles/pmc_workload.py > $OUT/$v.log 2>&1
done
cp /tmp/keep_full.so $R/ssa-gym_amd/libssa_hip.so
cd $R
for v in full a1 a2 a4 a8 a16 a32; do echo "== $v"; python3 profiles/pmc_counters_reduce.py gpurun_out/ablate/$v | python3 -c "
import json,sys
d=json.load(sys.stdin)['counters']
print({k: round(v['per_wavefront'],1) for k,v in d.items() if isinstance(v,dict)})"; done
