# round 2, run X: cooperative "bits" reduction levels from the first level on (shorter dependency chain) vs lane-serial wide levels
set -x
mkdir -p gpurun_out
make -s -C oracle
B20="python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0"
B24="python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0"
PC_HIP_COOP_MAX_LOG2=22 timeout -k 10 900 python -m pytest tests/test_msm_gpu.py tests/test_kzg_gpu.py -q -x 2>&1 | tail -2
for v in 17 19 22; do
  PC_HIP_COOP_MAX_LOG2=$v timeout -k 10 600 $B20 > gpurun_out/x_2p20_c$v.json 2>/dev/null
  PC_HIP_COOP_MAX_LOG2=$v timeout -k 10 600 $B20 --inflight 0 > gpurun_out/x_2p20_blocking_c$v.json 2>/dev/null
done
for v in 17 22; do
  PC_HIP_COOP_MAX_LOG2=$v timeout -k 10 600 $B24 > gpurun_out/x_2p24_c$v.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/x_*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d["steps"], round(d.get("blocking_msm_ms"),3), {k: round(v,3) for k,v in d.get("msm_phase_ms").items()})
    except Exception as e: print(f, "failed", e)
PY
