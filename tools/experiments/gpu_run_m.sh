# round 2, run M: field products as calls (instruction-cache hypothesis); Hyrax C++ mirror test
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 600 python -m pytest tests/test_hyrax_gpu.py -q -x -k cpp > gpurun_out/m_pytest_hyrax.log 2>&1; tail -5 gpurun_out/m_pytest_hyrax.log
B="python bench.py --no-cpu-baseline --no-h2d"
timeout -k 10 600 $B > gpurun_out/m_base.json 2>/dev/null
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_calls.so timeout -k 10 600 $B > gpurun_out/m_calls.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/m_*.json")):
    try:
        d = json.load(open(f))
        s = d.get("secondary") or {}
        print(f, round(d["ms_per_step"], 2), d.get("blocking_msm_ms"), {k: round(v, 2) for k, v in (d.get("msm_phase_ms") or {}).items()},
              "| 2^20", s.get("ms_per_step"), s.get("blocking_msm_ms"), (s.get("msm_phase_ms") or {}).get("accumulate"))
    except Exception as e: print(f, "failed", e)
PY
