#!/bin/bash
# static VALU/SALU/LDS instruction count of the COMMON path of step_fast_kernel<1,false> (update block, jitter ladder and the
# non-elliptic call compiled out in a scratch copy of the sources) -- a proxy for the per-wavefront dynamic count
set -e
SRC=${1:-/root/repo}
W=/tmp/isa/hot_$$; mkdir -p $W/ssa-gym_amd/csrc $W/include
cp $SRC/ssa-gym_amd/csrc/* $W/ssa-gym_amd/csrc/; cp $SRC/include/ssa_hip.h $W/include/
python3 - "$W" <<'PY'
import sys
w=sys.argv[1]
p=w+'/ssa-gym_amd/csrc/ssa_kernels.hip'; s=open(p).read()
a="const bool my_update = valid && act >= 0 && (int64_t)act == j && interval_ok;"
assert a in s; s=s.replace(a,"const bool my_update = false;")
a="    if (C.flags & SSA_FLAG_RESAMPLE) {\n        const int rg"
assert a in s; s=s.replace(a,"    if (false) {\n        const int rg")
open(p,'w').write(s)
p=w+'/ssa-gym_amd/csrc/ssa_math.hpp'; m=open(p).read()
a="if (__any(!handled)) {   // whole-wave branch"
if a in m: m=m.replace(a,"if (false) {")
open(p,'w').write(m)
PY
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=8 -gline-tables-only -S --cuda-device-only ${EXTRA} -o $W/hot.s $W/ssa-gym_amd/csrc/ssa_kernels.hip 2>/dev/null
python3 /root/repo/build_ablate/isa_by_line.py $W/hot.s '_ZN3ssa16step_fast_kernelILi1ELb0EEEviiPKdS2_S2_PKiNS_5StepKE' --ops > $W/lines.txt
tail -2 $W/lines.txt | head -1
cp $W/lines.txt /tmp/isa/hot_lines_latest.txt
grep -A3 "step_fast_kernelILi1ELb0EEEvNS_5StepKEii$" $W/hot.s | head -0
awk '/\.name: *_ZN3ssa16step_fast_kernelILi1ELb0EEEvii/{f=1} f&&/vgpr_count|sgpr_count|private_segment_fixed|vgpr_spill/{print} f&&/wavefront_size/{exit}' $W/hot.s
rm -rf $W
