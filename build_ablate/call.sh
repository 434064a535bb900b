mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r2z_pytest.log
timeout -k 10 300 python build_ablate/gym_profile.py > gpurun_out/r2z_gym.txt 2>&1; grep "====" gpurun_out/r2z_gym.txt
timeout -k 10 300 python - <<'PY'
import sys; sys.argv=['bench.py']
import bench
for mode in ('flatten','aer'):
    print(mode, bench.gym_api_rate(20000, mode))
for mode in ('flatten','aer'):
    print("2000 objects", mode, bench.gym_api_rate(2000, mode))
PY
