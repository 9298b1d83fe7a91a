mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for inf in 0 2; do
timeout -k 10 600 python bench.py --inflight $inf --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));print('inflight',$inf,d['value'],d['ms_per_step'],d['msm_phase_ms'])"
done
timeout -k 10 600 python bench.py --workload batch --steps 2 --warmup 1 2>/dev/null | cut -c100-330
timeout -k 10 900 python bench.py --log-degree 24 --steps 3 --warmup 1 --no-cpu-baseline --inflight 0 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));print('2^24',d['value'],d['ms_per_step'],d['msm_phase_ms'])"
