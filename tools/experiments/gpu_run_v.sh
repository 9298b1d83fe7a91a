# round 2, run V: HW queue priority of the reduction (tail) streams and of the main streams
set -x
mkdir -p gpurun_out
make -s -C oracle
B20="python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0"
B24="python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0"
run() { tag=$1; shift; env "$@" timeout -k 10 600 $B20 > gpurun_out/v_2p20_$tag.json 2>/dev/null; env "$@" timeout -k 10 600 $B24 > gpurun_out/v_2p24_$tag.json 2>/dev/null; env "$@" timeout -k 10 600 python bench.py --workload batch > gpurun_out/v_batch_$tag.json 2>/dev/null; }
run default X=1
run tailhi PC_HIP_TAIL_PRIO=hi
run tail0 PC_HIP_TAIL_PRIO=0
run tailhi_mainlo PC_HIP_TAIL_PRIO=hi PC_HIP_MAIN_PRIO=lo
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/v_*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d["steps"], d.get("blocking_msm_ms"))
    except Exception as e: print(f, "failed", e)
PY
