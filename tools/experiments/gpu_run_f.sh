set -x
mkdir -p gpurun_out
make -s -C oracle
if ! timeout -k 10 300 python -m pytest tests/test_msm_gpu.py -m gpu -q -x -k "window_table" > gpurun_out/f_sanity.log 2>&1; then
  tail -30 gpurun_out/f_sanity.log
  PC_HIP_TABLE_BUILD=serial timeout -k 10 300 python -m pytest tests/test_msm_gpu.py -m gpu -q -x -k "window_table" 2>&1 | tail -5
  exit 1
fi
tail -3 gpurun_out/f_sanity.log
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/f_pytest.log 2>&1; tail -8 gpurun_out/f_pytest.log
for fb in 8 9 10 11; do
  PC_HIP_FINE_BITS=$fb timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d > gpurun_out/f_fb$fb.json 2>/dev/null
  PC_HIP_FINE_BITS=$fb timeout -k 10 300 python bench.py --workload batch --steps 5 > gpurun_out/f_batch_fb$fb.json 2>/dev/null
done
python - <<'PY'
import json
for fb in (8, 9, 10, 11):
    d = json.load(open(f"gpurun_out/f_fb{fb}.json")); s = d["secondary"]; b = json.load(open(f"gpurun_out/f_batch_fb{fb}.json"))
    print(fb, "2^24", round(d["ms_per_step"], 2), round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()},
          "| 2^20", round(s["ms_per_step"], 2), round(s["blocking_msm_ms"], 2), {k: round(v, 3) for k, v in s["msm_phase_ms"].items()}, "| batch", round(b["ms_per_step"], 1))
PY
