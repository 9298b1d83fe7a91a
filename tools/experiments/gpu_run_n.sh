# round 2, run N: modulus limbs as SGPR operands
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 120 tools/microbench > gpurun_out/n_microbench.txt 2>&1; tail -9 gpurun_out/n_microbench.txt
timeout -k 10 900 python -m pytest tests/test_msm_gpu.py tests/test_ntt_gpu.py -q -x > gpurun_out/n_pytest.log 2>&1; tail -3 gpurun_out/n_pytest.log
B="python bench.py --no-cpu-baseline --no-h2d"
timeout -k 10 600 $B > gpurun_out/n_base.json 2>/dev/null
timeout -k 10 600 python bench.py --workload ntt > gpurun_out/n_ntt.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/n_*.json")):
    try:
        d = json.load(open(f))
        s = d.get("secondary") or {}
        print(f, round(d["ms_per_step"], 3), d.get("blocking_msm_ms"), {k: round(v, 2) for k, v in (d.get("msm_phase_ms") or {}).items()},
              "| 2^20", s.get("ms_per_step"), s.get("blocking_msm_ms"), (s.get("msm_phase_ms") or {}).get("accumulate"), d.get("ntt_phase_ms"), d.get("column_hash_blake2s_ms"))
    except Exception as e: print(f, "failed", e)
PY
