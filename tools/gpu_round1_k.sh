mkdir -p gpurun_out
make -s -C oracle
PC_BENCH_FORCE_DIST=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29777 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-400
