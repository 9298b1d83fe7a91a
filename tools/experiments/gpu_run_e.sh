set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/e_pytest.log 2>&1; tail -8 gpurun_out/e_pytest.log
timeout -k 10 600 python bench.py --workload batch > gpurun_out/e_bench_batch.json 2>/dev/null
PC_HIP_BATCH_G=0 timeout -k 10 600 python bench.py --workload batch > gpurun_out/e_bench_batch_g0.json 2>/dev/null
PC_HIP_BATCH_G=4 timeout -k 10 600 python bench.py --workload batch > gpurun_out/e_bench_batch_g4.json 2>/dev/null
PC_HIP_BATCH_G=16 timeout -k 10 600 python bench.py --workload batch > gpurun_out/e_bench_batch_g16.json 2>/dev/null
for f in gpurun_out/e_bench_batch*.json; do python -c "import json; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'])"; done
