mkdir -p gpurun_out
timeout -k 10 300 python build_ablate/closed_loop_parts.py > gpurun_out/r2z_cl.txt 2>&1; cat gpurun_out/r2z_cl.txt
