set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_host_split_gpu.py tests/test_external_vectors_gpu.py tests/test_msm_gpu.py tests/test_kzg_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout -k 10 900 python bench.py > gpurun_out/p4_bench.json 2> gpurun_out/p4_bench.err || tail -20 gpurun_out/p4_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/p4_bench.json")); s=d.get("secondary")
t=d["trait_shaped"]; ts=s["trait_shaped"]
print("2^24", round(d["ms_per_step"], 2), round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()}, d["parity"]["commit_ok"], d["parity"]["open_ok"])
print(" trait", round(t["ms_per_commit_open"],1), "commit", round(t["commit_ms"],1), "open", round(t["open_ms"],1), "cache", round(t["with_shim_polynomial_cache_ms"],1), t["parity_ok"], t["with_shim_polynomial_cache_parity_ok"], "h2d", round(d["value_h2d_inclusive"]["ms_per_step"],2))
print("2^20", round(s["ms_per_step"], 2), round(s["blocking_msm_ms"], 2), "trait", round(ts["ms_per_commit_open"],2), round(ts["with_shim_polynomial_cache_ms"],2), ts["parity_ok"])
w=d["workloads"]
print("batch", round(w["batch"]["ms_per_step"],2), json.dumps(w["batch"].get("roofline"))[:600])
print("ipa", round(w["ipa"]["commit_ms"],2), round(w["ipa"]["open_ms"],2), json.dumps(w["ipa"].get("roofline"))[:400])
print("ligero", w["ligero"]["ms_per_step"])
print("lat", {k:(round(v["gpu_commit_open_ms"],2), round(v.get("cpu_port_commit_open_ms",0),1)) for k,v in w["latency"]["rows"].items()})
print("cpu", json.dumps(d["cpu_baseline"])[:900])
print("wall", d["bench_wall_s"])
PY
