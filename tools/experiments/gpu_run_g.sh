set -x
mkdir -p gpurun_out
make -s -C oracle
if ! timeout -k 10 300 python -m pytest tests/test_msm_gpu.py -m gpu -q -x > gpurun_out/g_sanity.log 2>&1; then tail -30 gpurun_out/g_sanity.log; exit 1; fi
tail -2 gpurun_out/g_sanity.log
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d > gpurun_out/g_$tag.json 2>/dev/null; env "$@" timeout -k 10 300 python bench.py --workload batch --steps 5 > gpurun_out/g_batch_$tag.json 2>/dev/null; }
run default PC_X=0
run t256 PC_HIP_SORT_THREADS=256
run t512 PC_HIP_SORT_THREADS=512
run fb7 PC_HIP_FINE_BITS=7
run fb6 PC_HIP_FINE_BITS=6
python - <<'PY'
import json
for t in ("default", "t256", "t512", "fb7", "fb6"):
    try:
        d = json.load(open(f"gpurun_out/g_{t}.json")); s = d["secondary"]; b = json.load(open(f"gpurun_out/g_batch_{t}.json"))
        print(t, "2^24", round(d["ms_per_step"], 2), round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()},
              "| 2^20", round(s["ms_per_step"], 2), round(s["blocking_msm_ms"], 2), {k: round(v, 3) for k, v in s["msm_phase_ms"].items()}, "| batch", round(b["ms_per_step"], 1))
    except Exception as e:
        print(t, "failed", e)
PY
