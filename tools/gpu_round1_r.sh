mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_ntt_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 600 python bench.py --workload ntt --steps 5 --warmup 1 2>/dev/null > gpurun_out/bench_r01_ntt_c.json; python -c "
import json;d=json.load(open('gpurun_out/bench_r01_ntt_c.json'));print('ntt',d['value'],d['ms_per_step'],d['ntt_phase_ms'],d['roofline']['frac'])"
