make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_ipa_gpu.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -40
