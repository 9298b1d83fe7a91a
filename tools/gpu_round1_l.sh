mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python bench.py --workload batch --steps 2 --warmup 1 > gpurun_out/bench_r01_batch.json 2>gpurun_out/batch.err; tail -3 gpurun_out/batch.err; cat gpurun_out/bench_r01_batch.json
