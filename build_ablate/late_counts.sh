#!/bin/bash
# instruction counts of the step kernel per launch over one episode (PROP=hybrid|elements|fg): healthy (steps 60-180) against late (360-470)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=${PROP:-hybrid}
O=$R/gpurun_out/late_$P; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PROP=$P rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $O -- python3 $R/build_ablate/episode_profile.py > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("$O/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "step_fast_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
def mean(lo, hi, k):
    v = [rows[i][k] / rows[i]["SQ_WAVES"] for i in ids[lo:hi]]
    return sum(v) / len(v)
print("$P: %d launches" % len(ids))
for name, lo, hi in (("healthy (steps 60-180)", 60, 180), ("late (steps 360-470)", 360, 470)):
    print("  %-24s per wavefront: VALU %7.1f  SALU %7.1f  SMEM %5.1f  wave cycles %8.1f  VALU active %7.1f  waiting to issue %8.1f" % (
        name, mean(lo, hi, "SQ_INSTS_VALU"), mean(lo, hi, "SQ_INSTS_SALU"), mean(lo, hi, "SQ_INSTS_SMEM"), mean(lo, hi, "SQ_WAVE_CYCLES"),
        mean(lo, hi, "SQ_ACTIVE_INST_VALU"), mean(lo, hi, "SQ_WAIT_INST_ANY")))
PY
