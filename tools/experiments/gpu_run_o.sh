# round 2, run O: accumulate at three waves per SIMD without the register prefetch
set -x
mkdir -p gpurun_out
make -s -C oracle
B="python bench.py --no-cpu-baseline --no-h2d"
timeout -k 10 600 $B > gpurun_out/o_base.json 2>/dev/null
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_np3.so timeout -k 10 600 $B > gpurun_out/o_np3.json 2>/dev/null
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_np3.so PC_HIP_TBL_LANES=196608 timeout -k 10 600 $B > gpurun_out/o_np3_l192k.json 2>/dev/null
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_np3.so timeout -k 10 600 python -m pytest tests/test_msm_gpu.py -q -x 2>&1 | tail -2
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/o_*.json")):
    try:
        d = json.load(open(f))
        s = d.get("secondary") or {}
        print(f, round(d["ms_per_step"], 3), d.get("blocking_msm_ms"), {k: round(v, 2) for k, v in (d.get("msm_phase_ms") or {}).items()},
              "| 2^20", s.get("ms_per_step"), s.get("blocking_msm_ms"), (s.get("msm_phase_ms") or {}).get("accumulate"))
    except Exception as e: print(f, "failed", e)
PY
