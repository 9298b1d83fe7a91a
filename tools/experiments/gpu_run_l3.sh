set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_ligero_gpu.py tests/test_ntt_gpu.py -q -x 2>&1 | tail -5
