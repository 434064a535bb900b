mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -s -rA > gpurun_out/r2z_pytest_verbose.log 2>&1; echo "pytest rc $?"; grep -c "PASSED" gpurun_out/r2z_pytest_verbose.log; grep "^\[" gpurun_out/r2z_pytest_verbose.log | cut -c1-260 | head -60
