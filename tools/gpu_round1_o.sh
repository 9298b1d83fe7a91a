mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for extra in "--inflight 0" "--inflight 2"; do
timeout -k 10 600 python bench.py $extra --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));print('$extra',d['value'],d['ms_per_step'],d['msm_phase_ms'])"
done
