mkdir -p gpurun_out
EPISODES=400 timeout -k 10 900 python build_ablate/soak.py > gpurun_out/r2z_soak.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids gpurun_out/r2z_soak.txt | cut -c1-400
