# round 2, run L: Hyrax harness tests, full GPU suite with the fused formulas, bench + exchange timers
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 600 python -m pytest tests/test_hyrax_gpu.py -q -x > gpurun_out/l_pytest_hyrax.log 2>&1; tail -15 gpurun_out/l_pytest_hyrax.log
timeout -k 10 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_hyrax_gpu.py > gpurun_out/l_pytest.log 2>&1; tail -5 gpurun_out/l_pytest.log
B="python bench.py --no-cpu-baseline"
timeout -k 10 600 $B > gpurun_out/l_base.json 2>/dev/null
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
PC_BENCH_FORCE_DIST=1 timeout -k 10 600 $B --secondary-log-degree 0 --no-h2d > gpurun_out/l_dist_kzg.json 2> gpurun_out/l_dist_kzg.err; tail -2 gpurun_out/l_dist_kzg.err
unset MASTER_ADDR MASTER_PORT RANK LOCAL_RANK WORLD_SIZE
timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/l_ipa_2p22.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/l_*.json")):
    try:
        d = json.load(open(f))
        if "ms_per_step" not in d: print(f, {k: d[k] for k in ("commit_ms", "open_rounds_ms") if k in d}); continue
        s = d.get("secondary") or {}
        print(f, round(d["ms_per_step"], 2), d.get("blocking_msm_ms"), {k: round(v, 2) for k, v in (d.get("msm_phase_ms") or {}).items()},
              "| 2^20", s.get("ms_per_step"), s.get("blocking_msm_ms"), ((d.get("roofline") or {}).get("arithmetic") or {}).get("frac"), d.get("exchange_host_ms"))
    except Exception as e: print(f, "failed", e)
PY
