make -s -C oracle
timeout -k 5 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for inf in 0 2; do
timeout -k 5 90 python bench.py --inflight $inf --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));print('inflight',$inf,d['value'],d['ms_per_step'],d['msm_phase_ms'])"
done
