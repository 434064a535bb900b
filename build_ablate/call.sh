mkdir -p gpurun_out/skip
cp ssa-gym_amd/libssa_hip.so /tmp/keep.so
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/skip
export PROP=fg M=160000
for v in full s1 s2 s4 s8 s31; do
  if [ $v = full ]; then cp /tmp/keep.so $R/ssa-gym_amd/libssa_hip.so; else cp $R/build_ablate/skip/$v.so $R/ssa-gym_amd/libssa_hip.so; fi
  timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$v -- python3 $R/profiles/pmc_workload.py > $OUT/$v.log 2>&1 || { echo "rocprof failed for $v"; break; }
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/$v/*/*counter_collection.csv")[0]
acc={}
for r in csv.DictReader(open(f)):
    if 'step_fast_kernel' in r['Kernel_Name']: acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
print("$v", {k: round(sum(v[20:])/len(v[20:])*1024/1e6,1) for k,v in acc.items()}, "MB per launch (raw counter)")
PY
done
cp /tmp/keep.so $R/ssa-gym_amd/libssa_hip.so
