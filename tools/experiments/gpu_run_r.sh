# round 2, run R: results in flight (queue depth) at 2^20 and 2^24
set -x
mkdir -p gpurun_out
make -s -C oracle
for d in 2 3 4 6; do
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0 --inflight $d > gpurun_out/r_2p20_d$d.json 2>/dev/null
done
for d in 3 4; do
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0 --inflight $d > gpurun_out/r_2p24_d$d.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r_*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d["steps"], d.get("blocking_msm_ms"))
    except Exception as e: print(f, "failed", e)
PY
