mkdir -p gpurun_out
FAST=1 PROPS=fg timeout -k 10 400 python build_ablate/time_variants.py > gpurun_out/r2z_variants.txt 2>&1 ; cat gpurun_out/r2z_variants.txt
FAST=1 M=2000 PROPS=fg timeout -k 10 400 python build_ablate/time_variants.py > gpurun_out/r2z_variants2k.txt 2>&1 ; cat gpurun_out/r2z_variants2k.txt
