# final state check: whole GPU suite, smoke, default bench line
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > gpurun_out/y_pytest.log 2>&1; tail -3 gpurun_out/y_pytest.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/y_bench.json 2> gpurun_out/y_bench.err; tail -2 gpurun_out/y_bench.err; cut -c1-400 gpurun_out/y_bench.json
timeout -k 10 300 python tools/hyrax_timing.py 2>/dev/null | grep workload | cut -c1-220
