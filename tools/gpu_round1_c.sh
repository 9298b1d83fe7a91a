set -x
mkdir -p gpurun_out
./tools/microbench 2>&1 | tail -5
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r01_b.json 2> gpurun_out/bench_r01_b.err; tail -3 gpurun_out/bench_r01_b.err; cat gpurun_out/bench_r01_b.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_b.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R; find gpurun_out -name "*.csv" | head -20; du -sh gpurun_out
