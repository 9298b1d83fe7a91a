# round 2, run K: fused a*b+c*d in the mixed addition; fused per-step collective
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 120 tools/microbench > gpurun_out/k_microbench.txt 2>&1; tail -8 gpurun_out/k_microbench.txt
timeout -k 10 900 python -m pytest tests/test_msm_gpu.py tests/test_kzg_gpu.py -q -x > gpurun_out/k_pytest.log 2>&1; tail -3 gpurun_out/k_pytest.log
B="python bench.py --no-cpu-baseline"
timeout -k 10 600 $B > gpurun_out/k_base.json 2>/dev/null
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
PC_BENCH_FORCE_DIST=1 timeout -k 10 600 $B --secondary-log-degree 0 --no-h2d > gpurun_out/k_dist_kzg.json 2> gpurun_out/k_dist_kzg.err; tail -2 gpurun_out/k_dist_kzg.err
unset MASTER_ADDR MASTER_PORT RANK LOCAL_RANK WORLD_SIZE
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/k_*.json")):
    try:
        d = json.load(open(f))
        s = d.get("secondary") or {}
        print(f, round(d["ms_per_step"], 2), d.get("blocking_msm_ms"), {k: round(v, 2) for k, v in (d.get("msm_phase_ms") or {}).items()},
              "| 2^20", s.get("ms_per_step"), s.get("blocking_msm_ms"), ((d.get("roofline") or {}).get("arithmetic") or {}).get("frac"))
    except Exception as e: print(f, "failed", e)
PY
