set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_host_cpp.py -q -x -m gpu 2>&1 | tail -12
