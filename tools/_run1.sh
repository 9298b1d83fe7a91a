timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PC_IPA_REPS=3 timeout 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 | cut -c1-900
