#!/bin/bash
# quick PMC pass: instruction mix of the step kernel (VALU / SALU / LDS / VMEM per wavefront)   bash build_ablate/quick_counters.sh tag
set -u
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/qc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_sq2.log 2>&1
cd $R
python3 profiles/pmc_counters_reduce.py gpurun_out/qc_$TAG/pmc_sq2 > $OUT/counters.json 2> $OUT/counters.err
python3 - <<PY
import json
d=json.load(open("$OUT/counters.json"))["counters"]
print({k: round(v["per_wavefront"],1) for k,v in d.items() if isinstance(v, dict)})
PY
