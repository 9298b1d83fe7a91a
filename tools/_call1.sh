set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 300 tools/microbench > gpurun_out/c1_microbench.txt 2>&1
timeout -k 10 1500 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/c1_pytest.log 2>&1; tail -15 gpurun_out/c1_pytest.log
VARIANTS="default noswz nop0" RUNS="ntt" PROF="ntt" SQ="ntt" LDSC="ntt" bash tools/gpu_probe.sh c1 > gpurun_out/c1_probe.log 2>&1
VARIANTS="noswz" RUNS="" LDSC="ntt" bash tools/gpu_probe.sh c1b > gpurun_out/c1b_probe.log 2>&1
VARIANTS="default nop0" RUNS="n8 ipa" bash tools/gpu_probe.sh c1c > gpurun_out/c1c_probe.log 2>&1
timeout -k 10 600 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; tail -3 gpurun_out/c1_bench.err
grep -v "^+" gpurun_out/c1_probe.log | tail -40
grep -v "^+" gpurun_out/c1b_probe.log | tail -8
grep -v "^+" gpurun_out/c1c_probe.log | tail -12
cat gpurun_out/c1_microbench.txt | head -30
