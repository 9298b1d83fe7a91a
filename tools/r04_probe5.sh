set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_glv_table_gpu.py tests/test_residency_gpu.py tests/test_msm_gpu.py -m gpu -x -q 2>&1 | tail -8
for g in 0 1; do
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --workloads none --glv-table $g > gpurun_out/p5_bench_glv$g.json 2> gpurun_out/p5_bench_glv$g.err || tail -5 gpurun_out/p5_bench_glv$g.err
done
timeout -k 10 600 python bench.py --workload batch --steps 3 --glv-table 1 > gpurun_out/p5_batch_glv1.json 2>/dev/null
PC_HIP_TBL_PAD=0 timeout -k 10 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --workloads none --glv-table 0 --secondary-log-degree 0 > gpurun_out/p5_bench_nopad.json 2>/dev/null
python - <<'PY'
import json
for f in ("p5_bench_glv0","p5_bench_glv1","p5_bench_nopad"):
    d=json.load(open(f"gpurun_out/{f}.json")); s=d.get("secondary")
    print(f, "2^24", round(d["ms_per_step"], 2), round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()}, d["parity"]["commit_ok"], d["parity"]["open_ok"], "tbl build", round(d["config"]["srs_window_table_build_ms"]), "trait", round(d["trait_shaped"]["ms_per_commit_open"],1), d["roofline"]["arithmetic"]["digits_per_scalar"], d["roofline"]["arithmetic"]["buckets"])
    if s: print("   2^20", round(s["ms_per_step"], 2), round(s["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in s["msm_phase_ms"].items()}, s["parity"]["commit_ok"], s["parity"]["open_ok"])
d=json.load(open("gpurun_out/p5_batch_glv1.json")); print("batch glv", d["ms_per_step"], d["parity"]["all_commitments_closed_form_ok"])
PY
