#!/bin/bash
# Collects the round's profiling evidence on the MI355X box (gpurun):  bash profiles/collect.sh r02
# kernel-trace statistics of the bench command, the two HBM-traffic PMC passes (+ calibration kernels), the instruction-mix and
# wait counters of the step kernel.  Counters are collected in their own passes with --kernel-trace only.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 958 --warmup 0 --no-cpu-baseline --no-legs --rollout 60 > $OUT/prof_bench.json 2> $OUT/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_f64 -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_f64.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_mfma.log 2>&1
cd $R
python3 profiles/pmc_reduce.py gpurun_out/$TAG/pmc_FETCH_SIZE gpurun_out/$TAG/pmc_WRITE_SIZE > $OUT/traffic.json 2> $OUT/traffic.err
python3 profiles/pmc_counters_reduce.py gpurun_out/$TAG/pmc_f64 gpurun_out/$TAG/pmc_sq1 gpurun_out/$TAG/pmc_sq2 gpurun_out/$TAG/pmc_mfma > $OUT/counters.json 2> $OUT/counters.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -8 $OUT/kernel_stats.csv
cat $OUT/traffic.json | tail -5
cat $OUT/counters.json | head -60
