mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 900 python tools/ipa_timing.py 22 2>/dev/null | tail -1
timeout -k 10 600 python bench.py --no-cpu-baseline 2>/dev/null | cut -c100-250
