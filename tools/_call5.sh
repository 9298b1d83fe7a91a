set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_msm_gpu.py tests/test_kzg_gpu.py tests/test_host_split_gpu.py -m gpu -q -x > gpurun_out/c5_pytest.log 2>&1; tail -3 gpurun_out/c5_pytest.log
for P in "1,3,4,8" "1,2,5,8" "1,2,4,9" "1,2,3,4,6"; do
  PC_HIP_HOST_PARTS=$P timeout -k 10 300 python tools/host_parts_probe.py 24 2>/dev/null | tail -1
done
VARIANTS="default" RUNS="kzg24 kzg20 n8" bash tools/gpu_probe.sh c5 > gpurun_out/c5a.log 2>&1
KZG_FLAGS="--inflight 0" VARIANTS="default" RUNS="kzg24" bash tools/gpu_probe.sh c5blk > gpurun_out/c5b.log 2>&1
grep "^==" gpurun_out/c5a.log gpurun_out/c5b.log | cut -c1-900
