set -x
mkdir -p gpurun_out
./tools/microbench > gpurun_out/microbench.txt 2>&1; cat gpurun_out/microbench.txt
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q 2>&1 | tail -30
