mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r2z_pytest.log
FAST=1 PROPS=fg timeout -k 10 400 python build_ablate/time_variants.py > gpurun_out/r2z_variants.txt 2>&1 ; cat gpurun_out/r2z_variants.txt
FAST=1 M=2000 PROPS=fg timeout -k 10 400 python build_ablate/time_variants.py > gpurun_out/r2z_variants2k.txt 2>&1 ; cat gpurun_out/r2z_variants2k.txt
