set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
for L in 131072 262144 524288; do
  PC_HIP_TBL_LANES=$L timeout -k 10 400 python tools/n8_probe.py > gpurun_out/p10_probe_$L.json 2> gpurun_out/p10_probe_$L.err || tail -3 gpurun_out/p10_probe_$L.err
  PC_HIP_TBL_LANES=$L timeout -k 10 400 python bench.py --steps 6 --no-cpu-baseline --workloads none --no-h2d --no-trait > gpurun_out/p10_bench_$L.json 2>/dev/null
done
python - <<'PY'
import json
for L in (131072, 262144, 524288):
    d=json.load(open(f"gpurun_out/p10_probe_{L}.json"))
    print(L, "bn254 single", d['bn254_single']['blocking_ms'], d['bn254_single']['phases_ms'], round(d['bn254_single']['accumulate_madd_per_s']/1e9,2), "| batch", round(d['bn254_batch']['ms_per_step'],2), "| pallas", d['pallas_single']['blocking_ms'], d['pallas_single']['phases_ms'], round(d['pallas_single']['accumulate_madd_per_s']/1e9,2), d['bn254_single']['parity_ok'], d['bn254_batch']['parity_ok'], d['pallas_single']['parity_ok'])
    d=json.load(open(f"gpurun_out/p10_bench_{L}.json")); s=d["secondary"]
    print(L, "bls 2^24 step", round(d["ms_per_step"],2), "blocking", round(d["blocking_msm_ms"],2), {k: round(x, 2) for k, x in d["msm_phase_ms"].items()}, "| 2^20", round(s["ms_per_step"],2), round(s["blocking_msm_ms"],2), {k: round(x, 2) for k, x in s["msm_phase_ms"].items()}, d["parity"]["commit_ok"], s["parity"]["open_ok"])
PY
