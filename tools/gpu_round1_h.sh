set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 600 python bench.py --workload ntt --steps 5 --warmup 1 > gpurun_out/bench_r01_ntt_b.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/bench_r01_ntt_b.json'));print('ntt',d['value'],d['ms_per_step'],d['ntt_phase_ms'],d['roofline']['frac'])"
timeout -k 10 600 python bench.py > gpurun_out/bench_r01_f.json 2>/dev/null; cat gpurun_out/bench_r01_f.json
timeout -k 10 900 python bench.py --log-degree 24 --steps 3 --warmup 1 > gpurun_out/bench_r01_2p24_c.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/bench_r01_2p24_c.json'));print('2^24',d['value'],d['ms_per_step'],d['msm_phase_ms'],d['cpu_baseline'])"
