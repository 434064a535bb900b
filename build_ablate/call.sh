mkdir -p gpurun_out
timeout -k 10 600 python - > gpurun_out/r2z_vec.txt 2>&1 <<'PY'
import sys; sys.argv=['bench.py']
import bench
print(bench.vec_env_rate(20000))
PY
tail -3 gpurun_out/r2z_vec.txt
