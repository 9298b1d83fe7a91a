set -x
mkdir -p gpurun_out
make -s -C oracle
B="python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0 --steps 20"
timeout -k 10 600 $B > gpurun_out/p2_base20.json 2>/dev/null
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29536 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
PC_BENCH_FORCE_DIST=1 timeout -k 10 600 $B > gpurun_out/p2_dist.json 2>/dev/null
PC_BENCH_FORCE_DIST=1 PC_DIAG_DIST_NO_EXCHANGE=1 timeout -k 10 600 $B > gpurun_out/p2_dist_noexchange.json 2>/dev/null
unset MASTER_ADDR MASTER_PORT RANK LOCAL_RANK WORLD_SIZE
timeout -k 10 600 $B > gpurun_out/p2_base20b.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/p2_*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 3), d["steps"], d.get("blocking_msm_ms"), d.get("exchange_host_ms"))
    except Exception as e: print(f, "failed", e)
PY
