set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 400 python tools/n8_probe.py > gpurun_out/p3_probe.json 2> gpurun_out/p3_probe.err || tail -5 gpurun_out/p3_probe.err
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --workloads none > gpurun_out/p3_bench.json 2> gpurun_out/p3_bench.err || tail -5 gpurun_out/p3_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/p3_probe.json"))
print("bn254 single", round(d['bn254_single']['blocking_ms'],3), d['bn254_single']['phases_ms'], round(d['bn254_single']['accumulate_madd_per_s']/1e9,2), "| batch", round(d['bn254_batch']['ms_per_step'],2), "| pallas", round(d['pallas_single']['blocking_ms'],3), d['pallas_single']['phases_ms'], round(d['pallas_single']['accumulate_madd_per_s']/1e9,2), d['bn254_single']['parity_ok'], d['bn254_batch']['parity_ok'], d['pallas_single']['parity_ok'])
d=json.load(open("gpurun_out/p3_bench.json")); s=d.get("secondary")
print("2^24", round(d["ms_per_step"], 2), round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()}, d["parity"]["commit_ok"], d["parity"]["open_ok"], "trait", round(d["trait_shaped"]["ms_per_commit_open"],1), round(d["trait_shaped"]["with_shim_polynomial_cache_ms"],1), "h2d", round(d["value_h2d_inclusive"]["ms_per_step"],2))
print("2^20", round(s["ms_per_step"], 2), round(s["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in s["msm_phase_ms"].items()}, s["parity"]["commit_ok"], s["parity"]["open_ok"])
print("arith", d["roofline"]["arithmetic"]["frac"], d["roofline"]["kernel_ms"])
PY
