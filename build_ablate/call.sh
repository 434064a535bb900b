mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r2z_pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r2z_pytest.log
MASTER_ADDR=127.0.0.1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 build_ablate/sharded_host.py > gpurun_out/r2z_shost.txt 2>&1; echo rc $?; grep "overlap" gpurun_out/r2z_shost.txt | tail -8
