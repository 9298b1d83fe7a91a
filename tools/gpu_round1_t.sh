mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_msm_gpu.py tests/test_kzg_gpu.py -m gpu -x -q 2>&1 | tail -3
for inf in 0 2 3; do
timeout -k 10 600 python bench.py --inflight $inf --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));print('inflight',$inf,d['value'],d['ms_per_step'],d['msm_phase_ms'])"
done
timeout -k 10 600 python bench.py --workload batch --steps 2 --warmup 1 2>/dev/null | cut -c100-330
