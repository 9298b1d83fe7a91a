set -x
mkdir -p gpurun_out
make -s -C oracle
for rep in 1 2; do for g in 0 1; do
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workloads none --no-h2d --secondary-log-degree 0 --glv-table $g > gpurun_out/p6_glv${g}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json
for rep in (1,2):
  for g in (0,1):
    d=json.load(open(f"gpurun_out/p6_glv{g}_{rep}.json"))
    print("glv",g,"rep",rep, "step", round(d["ms_per_step"], 2), "blocking", round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()}, d["parity"]["commit_ok"], d["parity"]["open_ok"], "tbl build", round(d["config"]["srs_window_table_build_ms"]), "trait", round(d["trait_shaped"]["ms_per_commit_open"],1), round(d["trait_shaped"]["with_shim_polynomial_cache_ms"],1))
PY
