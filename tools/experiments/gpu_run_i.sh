set -x
mkdir -p gpurun_out
make -s -C oracle
B="python bench.py --secondary-log-degree 0 --no-h2d --no-cpu-baseline"
for rep in 1 2 3; do
  timeout -k 10 300 $B > gpurun_out/i_l17_$rep.json 2>/dev/null
  PC_HIP_TBL_LANES=262144 timeout -k 10 300 $B > gpurun_out/i_l18_$rep.json 2>/dev/null
  PC_HIP_TBL_LANES=524288 timeout -k 10 300 $B > gpurun_out/i_l19_$rep.json 2>/dev/null
done
PC_HIP_TBL_PAD=1 timeout -k 10 300 $B > gpurun_out/i_pad_1.json 2>/dev/null
PC_HIP_TBL_PAD=1 timeout -k 10 300 $B > gpurun_out/i_pad_2.json 2>/dev/null
PC_HIP_TBL_PAD=1 PC_HIP_TBL_LANES=524288 timeout -k 10 300 $B > gpurun_out/i_pad19_1.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/i_*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 2), round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()})
    except Exception as e: print(f, "failed", e)
PY
