cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_host_split_gpu.py tests/test_kzg_gpu.py tests/test_ntt_gpu.py tests/test_group_gpu.py -m gpu -q -x 2>&1 | tail -3
for lg in 20 21 22 23 24; do
  echo "lg=$lg"; timeout -k 10 300 python tools/host_parts_probe.py $lg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('resident_msm_ms','host_commit_ms','resident_open_ms','host_open_ms','parity')})"
done
for P in "1,3" "3,1"; do echo "parts=$P"; PC_HIP_HOST_PARTS=$P timeout -k 10 300 python tools/host_parts_probe.py 22 2>&1 | tail -1 | cut -c1-300; done
VARIANTS="default" RUNS="ntt" bash tools/gpu_probe.sh c8 > gpurun_out/c8a.log 2>&1; grep "^==" gpurun_out/c8a.log
