mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "multi_tile_wavefronts" > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r2z_pytest.log
