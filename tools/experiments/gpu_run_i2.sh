set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_ipa_gpu.py -q -x 2>&1 | tail -4
timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 | tee gpurun_out/i2_ipa.json | cut -c1-600
