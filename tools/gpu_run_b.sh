# GPU call B of round 2: suite after IPA / group / sqr changes, microbench, bench, IPA timing, PMC traffic passes.
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 120 tools/microbench > gpurun_out/b_microbench.txt 2>&1; cat gpurun_out/b_microbench.txt
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/b_pytest.log 2>&1; tail -20 gpurun_out/b_pytest.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/b_ipa_2p22.json; cat gpurun_out/b_ipa_2p22.json
timeout -k 10 900 python bench.py > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; tail -3 gpurun_out/b_bench.err
timeout -k 10 600 python bench.py --workload ntt > gpurun_out/b_bench_ntt.json 2>/dev/null
timeout -k 10 600 python bench.py --workload batch > gpurun_out/b_bench_batch.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B24="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --inflight 0 --secondary-log-degree 0"
B20="python $R/bench.py --log-degree 20 --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --inflight 0 --secondary-log-degree 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b_prof20 -o bench -- $B20 > $R/gpurun_out/b_prof20.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/b_fetch24 -o bench -- $B24 > $R/gpurun_out/b_fetch24.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/b_write24 -o bench -- $B24 > $R/gpurun_out/b_write24.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/b_fetch20 -o bench -- $B20 > $R/gpurun_out/b_fetch20.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/b_write20 -o bench -- $B20 > $R/gpurun_out/b_write20.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/b_fetchntt -o bench -- python $R/bench.py --workload ntt --steps 2 --warmup 1 > $R/gpurun_out/b_fetchntt.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/b_writentt -o bench -- python $R/bench.py --workload ntt --steps 2 --warmup 1 > $R/gpurun_out/b_writentt.log 2>&1
cd $R
find gpurun_out -name "*.csv" -size +20M -delete 2>/dev/null
head -c 600 gpurun_out/b_bench.json; echo; cat gpurun_out/b_bench_ntt.json | head -c 600; echo; head -c 500 gpurun_out/b_bench_batch.json
