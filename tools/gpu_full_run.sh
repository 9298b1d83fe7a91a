set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/run_bench_n1.json 2>/dev/null
timeout -k 10 600 python bench.py --inflight 0 --no-cpu-baseline > gpurun_out/run_bench_n1_inflight0.json 2>/dev/null
timeout -k 10 600 python bench.py --precompute 0 --no-cpu-baseline > gpurun_out/run_bench_n1_notable.json 2>/dev/null
timeout -k 10 600 python bench.py --precompute 0 --inflight 0 --no-cpu-baseline > gpurun_out/run_bench_n1_notable_inflight0.json 2>/dev/null
timeout -k 10 300 python bench.py --log-degree 22 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/run_bench_2p22.json 2>/dev/null
timeout -k 10 200 python tools/lincomb_timing.py 2>/dev/null | tail -1 > gpurun_out/run_lincomb.json
timeout -k 10 600 python bench.py --workload ntt --steps 5 --warmup 1 > gpurun_out/run_bench_ntt.json 2>/dev/null
timeout -k 10 600 python bench.py --workload batch --steps 2 --warmup 1 > gpurun_out/run_bench_batch.json 2>/dev/null
timeout -k 10 300 python bench.py --log-degree 24 --steps 3 --warmup 2 > gpurun_out/run_bench_2p24.json 2>/dev/null
timeout -k 10 200 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/run_ipa_2p22.json
timeout -k 10 120 tools/microbench > gpurun_out/run_microbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/run_prof -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --inflight 0 > $R/gpurun_out/run_prof.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/run_prof_notable -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --inflight 0 --precompute 0 > $R/gpurun_out/run_prof_notable.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/run_pmc_fetch -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 0 > $R/gpurun_out/run_pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/run_pmc_write -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 0 > $R/gpurun_out/run_pmc_write.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/run_prof_ntt -o bench -- python $R/bench.py --workload ntt --steps 3 --warmup 1 > $R/gpurun_out/run_prof_ntt.log 2>&1
cd $R; for f in gpurun_out/run_*.json; do echo $f; cut -c1-400 $f; echo; done
