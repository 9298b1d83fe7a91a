set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout -k 10 600 python tools/ipa_timing.py 20 > gpurun_out/ipa_r01_2p20.json 2> gpurun_out/ipa.err; tail -2 gpurun_out/ipa.err; cat gpurun_out/ipa_r01_2p20.json
timeout -k 10 900 python tools/ipa_timing.py 22 > gpurun_out/ipa_r01_2p22.json 2> gpurun_out/ipa.err; tail -2 gpurun_out/ipa.err; cat gpurun_out/ipa_r01_2p22.json
