#!/bin/bash
# per-stage dynamic instruction counts of the step kernel: PMC passes over the truncation builds, differenced stage by stage
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/trunc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$LATE" ]; then export STATE=/tmp/late_state.npz; GEN=$LATE LIB=$R/ssa-gym_amd/libssa_hip.so python3 $R/build_ablate/trunc_workload.py; export TICK0=$LATE; fi
for v in t1 t2 t3 t4 t5 t6 t7 t8 t9 full; do
  if [ $v = full ]; then export LIB=$R/ssa-gym_amd/libssa_hip.so; else export LIB=$R/build_ablate/trunc/$v.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/${v}_a -- python3 $R/build_ablate/trunc_workload.py > $OUT/${v}_a.log 2>&1 || echo "pass a failed for $v"
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/${v}_b -- python3 $R/build_ablate/trunc_workload.py > $OUT/${v}_b.log 2>&1 || echo "pass b failed for $v"
  echo "done $v"
done
cd $R
python3 build_ablate/trunc_reduce.py gpurun_out/trunc | tee gpurun_out/trunc_stages.txt
