mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2z_pytest.log
