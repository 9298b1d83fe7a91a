# GPU call A of round 2: full GPU suite (incl. BASELINE-size parity), default bench, kernel traces at 2^24 / 2^20, SQ counters.
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/a_pytest.log 2>&1; tail -25 gpurun_out/a_pytest.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; tail -3 gpurun_out/a_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/a_prof24 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --inflight 0 --secondary-log-degree 0 > $R/gpurun_out/a_prof24.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/a_prof20 -o bench -- python $R/bench.py --log-degree 20 --steps 5 --warmup 1 --no-cpu-baseline --no-h2d --inflight 0 --secondary-log-degree 0 > $R/gpurun_out/a_prof20.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/a_pmc_sq20 -o bench -- python $R/bench.py --log-degree 20 --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --inflight 0 --secondary-log-degree 0 > $R/gpurun_out/a_pmc_sq20.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/a_pmc_sqntt -o bench -- python $R/bench.py --workload ntt --steps 2 --warmup 1 > $R/gpurun_out/a_pmc_sqntt.log 2>&1
cd $R
ls gpurun_out/a_prof24 gpurun_out/a_prof20 gpurun_out/a_pmc_sq20 2>/dev/null | head -30
find gpurun_out/a_pmc_sq20 gpurun_out/a_pmc_sqntt -name "*.csv" -size +20M -delete 2>/dev/null
head -c 1500 gpurun_out/a_bench.json
