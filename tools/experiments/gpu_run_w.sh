# round 2, run W: wave priority only for the bodies that opt in (division scans, segmented / bucket reduction levels)
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 600 python bench.py --no-cpu-baseline --no-h2d > gpurun_out/w_default.json 2>/dev/null
timeout -k 10 600 python bench.py --no-cpu-baseline --no-h2d --precompute 0 > gpurun_out/w_notable.json 2>/dev/null
timeout -k 10 600 python bench.py --workload batch > gpurun_out/w_batch.json 2>/dev/null
timeout -k 10 300 python tools/hyrax_timing.py 2>/dev/null | grep workload > gpurun_out/w_hyrax.jsonl
cat gpurun_out/w_hyrax.jsonl
timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/w_ipa.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/w_*.json")):
    try:
        d = json.load(open(f))
        if "ms_per_step" not in d: print(f, d.get("commit_ms"), d.get("open_rounds_ms")); continue
        s = d.get("secondary") or {}
        print(f, round(d["ms_per_step"], 3), d["steps"], d.get("blocking_msm_ms"), "| 2^20", s.get("ms_per_step"), s.get("blocking_msm_ms"))
    except Exception as e: print(f, "failed", e)
PY
