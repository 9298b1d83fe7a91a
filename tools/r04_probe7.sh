set -x
mkdir -p gpurun_out
make -s -C oracle
for t in 1024 512 256; do
PC_HIP_SORT_THREADS=$t timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workloads none --no-h2d --secondary-log-degree 0 --glv-table 1 > gpurun_out/p7_glv_t$t.json 2>/dev/null
done
PC_BENCH_DEVICES=0,0,0,0,0,0,0,0 timeout -k 10 900 python bench.py --gpus 8 --log-degree 21 --no-cpu-baseline > gpurun_out/p7_8ranks.json 2> gpurun_out/p7_8ranks.err || tail -20 gpurun_out/p7_8ranks.err
python - <<'PY'
import json
for t in (1024,512,256):
    d=json.load(open(f"gpurun_out/p7_glv_t{t}.json"))
    print("glv threads",t, "step", round(d["ms_per_step"], 2), "blocking", round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()}, d["parity"]["commit_ok"], d["parity"]["open_ok"])
d=json.load(open("gpurun_out/p7_8ranks.json"))
print("8 ranks", d["n_gpus"], round(d["ms_per_step"],2), d["per_rank_ms_per_step"], d["parity"]["commit_ok"], d["parity"]["open_ok"], d["exchange_host_ms"], d["roofline"]["kernel_ms"])
PY
