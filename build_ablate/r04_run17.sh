#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_env_gpu.py -m gpu -x -q -k 'argmax or run_policy or agents' 2>&1 | tail -4
R=$(pwd); mkdir -p gpurun_out/r4q
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4q/prof -- python3 $R/build_ablate/argmax_probe.py > $R/gpurun_out/r4q/probe.log 2>&1
cd $R
find gpurun_out/r4q/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4q/kernel_stats.csv \;
head -8 gpurun_out/r4q/kernel_stats.csv | cut -c1-220
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r4q/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'masked_argmax' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
import numpy as np
d = np.array(d)
for k in range(0, len(d), 220):
    print(k, rows[k]['Kernel_Name'][:40], 'median ns', np.median(d[k:k+220]), 'wg', rows[k].get('Workgroup_Size_X'), rows[k].get('Grid_Size_X'), 'lds', rows[k].get('LDS_Block_Size'), 'scratch', rows[k].get('Scratch_Size'), 'vgpr', rows[k].get('VGPR_Count'))
PY
find gpurun_out/r4q/prof -name "*.csv" -size +5M -delete
