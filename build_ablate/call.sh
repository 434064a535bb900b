mkdir -p gpurun_out/xcd2
FAST=1 M=160000 PROPS=fg timeout -k 10 600 python build_ablate/time_variants.py > gpurun_out/r2z_variants160k.txt 2>&1 ; cat gpurun_out/r2z_variants160k.txt
cp ssa-gym_amd/libssa_hip.so /tmp/keep.so; cp build_ablate/v44_ntalways.so ssa-gym_amd/libssa_hip.so
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/xcd2
export PROP=fg M=160000
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_$c.log 2>&1; done
cd $R && python3 profiles/pmc_reduce.py gpurun_out/xcd2/pmc_FETCH_SIZE gpurun_out/xcd2/pmc_WRITE_SIZE > $OUT/traffic.json 2> $OUT/traffic.err; python3 -c "
import json; d=json.load(open('$OUT/traffic.json')); print('nt-always 160k', d.get('fg_160000'), (d.get('other_configurations') or {}).get('fg_160000',{}).get('step_raw'))"
cp /tmp/keep.so $R/ssa-gym_amd/libssa_hip.so
