set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
for P in "1,3,4" "4" "1,2,2,3" "1,3" "1,7" "1,3,4,8" "0"; do
  PC_HIP_HOST_PARTS=$P timeout -k 10 300 python tools/host_parts_probe.py 24 2>/dev/null | tail -1
done
PC_HIP_HOST_SPLIT_LOG2=10 PC_HIP_HOST_PARTS="1,3,4" timeout -k 10 300 python -m pytest tests/test_host_split_gpu.py -m gpu -q -x -k "parts4 or 4" 2>&1 | tail -2
