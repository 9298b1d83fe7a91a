set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_ntt_gpu.py tests/test_ligero_gpu.py tests/test_external_vectors_gpu.py -m gpu -q -x > gpurun_out/c6_pytest.log 2>&1; tail -3 gpurun_out/c6_pytest.log
VARIANTS="default" RUNS="ntt" bash tools/gpu_probe.sh c6 > gpurun_out/c6a.log 2>&1
grep "^==" gpurun_out/c6a.log
