# A/B on one box: library before the planner / graph changes vs HEAD, standalone 2^20 and 2^24
set -x
mkdir -p gpurun_out
make -s -C oracle
B20="python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0"
B24="python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0"
for rep in 1 2; do
  PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_old.so timeout -k 10 600 $B20 > gpurun_out/ab_2p20_old_$rep.json 2>/dev/null
  timeout -k 10 600 $B20 > gpurun_out/ab_2p20_head_$rep.json 2>/dev/null
  PC_HIP_GRAPHS=0 timeout -k 10 600 $B20 > gpurun_out/ab_2p20_head_nographs_$rep.json 2>/dev/null
done
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_old.so timeout -k 10 600 $B24 > gpurun_out/ab_2p24_old.json 2>/dev/null
timeout -k 10 600 $B24 > gpurun_out/ab_2p24_head.json 2>/dev/null
PC_HIP_GRAPHS=0 timeout -k 10 600 $B24 > gpurun_out/ab_2p24_head_nographs.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d["steps"], round(d.get("blocking_msm_ms"),3), {k: round(v,2) for k,v in d["msm_phase_ms"].items()})
    except Exception as e: print(f, "failed", e)
PY
