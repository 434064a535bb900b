mkdir -p gpurun_out
cp ssa-gym_amd/libssa_hip.so /tmp/keep.so; cp build_ablate/libs/trace.so ssa-gym_amd/libssa_hip.so
STEPS=380 timeout -k 10 200 python build_ablate/wave_timeline.py > gpurun_out/r2z_tl_late.txt 2>&1; grep -v "resident\|XCC\|distinct" gpurun_out/r2z_tl_late.txt | tail -40
cp /tmp/keep.so ssa-gym_amd/libssa_hip.so
