set -x
mkdir -p gpurun_out
make -s -C oracle
B="python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0 --steps 20"
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 600 $B > gpurun_out/q_base_q$q.json 2>/dev/null
done
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29536 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q PC_BENCH_FORCE_DIST=1 timeout -k 10 600 $B > gpurun_out/q_dist_q$q.json 2>/dev/null
done
unset MASTER_ADDR MASTER_PORT RANK LOCAL_RANK WORLD_SIZE
GPU_MAX_HW_QUEUES=8 timeout -k 10 600 python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0 > gpurun_out/q_2p20_q8.json 2>/dev/null
timeout -k 10 600 python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0 > gpurun_out/q_2p20_q4.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/q_*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 3), d["steps"], d.get("blocking_msm_ms"), d.get("exchange_host_ms"))
    except Exception as e: print(f, "failed", e)
PY
