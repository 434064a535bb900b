#!/bin/bash
# per-stage dynamic VALU count of the ELEMENTS chain (builds: see the loop in DESIGN / build below)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for L in $R/build_ablate/el/cut*.so $R/build_ablate/el/a_base.so; do
  T=$(basename $L .so); O=$R/gpurun_out/elc_$T; rm -rf $O; mkdir -p $O
  LIB=build_ablate/el/$T.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O -- python3 $R/build_ablate/el_chain_counts.py > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
  python3 - <<PY
import csv, glob
acc = {}
for f in glob.glob("$O/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "propagate_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
print("%-22s VALU per wavefront %8.1f   SALU %7.1f   (%d wavefronts, %d launches)" % ("$T", sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"]) / w, sum(acc["SQ_INSTS_SALU"]) / len(acc["SQ_INSTS_SALU"]) / w, w, len(acc["SQ_WAVES"])))
PY
done
