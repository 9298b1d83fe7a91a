set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r01_c.json 2> gpurun_out/bench_r01_c.err; tail -3 gpurun_out/bench_r01_c.err; cat gpurun_out/bench_r01_c.json
timeout -k 10 900 python bench.py --workload ntt --steps 5 --warmup 1 > gpurun_out/bench_r01_ntt.json 2> gpurun_out/bench_r01_ntt.err; tail -3 gpurun_out/bench_r01_ntt.err; cat gpurun_out/bench_r01_ntt.json
timeout -k 10 1200 python bench.py --log-degree 24 --steps 3 --warmup 1 > gpurun_out/bench_r01_2p24.json 2> gpurun_out/bench_r01_2p24.err; tail -3 gpurun_out/bench_r01_2p24.err; cat gpurun_out/bench_r01_2p24.json
