set -x
mkdir -p gpurun_out
make -s -C oracle
for t in 1024 512 256; do
PC_HIP_SORT_THREADS=$t timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workloads none --no-h2d --no-trait > gpurun_out/p12_t$t.json 2>/dev/null
done
python - <<'PY'
import json
for t in (1024,512,256):
    d=json.load(open(f"gpurun_out/p12_t{t}.json")); s=d["secondary"]
    print("sort threads",t, "2^24 step", round(d["ms_per_step"], 2), "blocking", round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()}, "| 2^20 step", round(s["ms_per_step"],2), round(s["blocking_msm_ms"],2), d["parity"]["commit_ok"], s["parity"]["open_ok"])
PY
