set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_glv_table_gpu.py tests/test_msm_gpu.py tests/test_kzg_gpu.py tests/test_residency_gpu.py -m gpu -q -x > gpurun_out/c2_pytest.log 2>&1; tail -4 gpurun_out/c2_pytest.log
timeout -k 10 600 python -m pytest tests/test_baseline_sizes_gpu.py -m gpu -q -x -k "kzg or bn254" > gpurun_out/c2_pytest2.log 2>&1; tail -4 gpurun_out/c2_pytest2.log
VARIANTS="default" RUNS="kzg24" bash tools/gpu_probe.sh c2glv > gpurun_out/c2a.log 2>&1
KZG_FLAGS="--glv-table 0" VARIANTS="default" RUNS="kzg24" bash tools/gpu_probe.sh c2full > gpurun_out/c2b.log 2>&1
KZG_FLAGS="--inflight 0" VARIANTS="default" RUNS="kzg24" bash tools/gpu_probe.sh c2glvblk > gpurun_out/c2c.log 2>&1
KZG_FLAGS="--inflight 0 --glv-table 0" VARIANTS="default" RUNS="kzg24" bash tools/gpu_probe.sh c2fullblk > gpurun_out/c2d.log 2>&1
KZG_FLAGS="--glv-table 1" VARIANTS="default" RUNS="kzg20" bash tools/gpu_probe.sh c2glv20 > gpurun_out/c2e.log 2>&1
VARIANTS="default" RUNS="kzg20" bash tools/gpu_probe.sh c2full20 > gpurun_out/c2f.log 2>&1
VARIANTS="default t256" RUNS="ntt" bash tools/gpu_probe.sh c2ntt > gpurun_out/c2g.log 2>&1
for f in a b c d e f g; do grep "^==" gpurun_out/c2$f.log; done
