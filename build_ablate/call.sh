mkdir -p gpurun_out
timeout -k 10 1000 bash profiles/collect.sh r02 > gpurun_out/r2z_collect.log 2>&1; echo "collect rc $?"; tail -30 gpurun_out/r2z_collect.log
