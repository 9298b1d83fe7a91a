set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_host_split_gpu.py tests/test_glv_table_gpu.py tests/test_kzg_gpu.py -m gpu -q -x > gpurun_out/c3_pytest.log 2>&1; tail -6 gpurun_out/c3_pytest.log
for P in 4 0 2 8; do
  PC_HIP_HOST_PARTS=$P timeout -k 10 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --workloads none --no-h2d --secondary-log-degree 0 > gpurun_out/c3_trait_$P.json 2> gpurun_out/c3_trait_$P.err || tail -5 gpurun_out/c3_trait_$P.err
done
KZG_FLAGS="--glv-table 1" VARIANTS="default" RUNS="kzg24" bash tools/gpu_probe.sh c3glv > gpurun_out/c3a.log 2>&1
KZG_FLAGS="--glv-table 1 --inflight 0" VARIANTS="default" RUNS="kzg24" bash tools/gpu_probe.sh c3glvblk > gpurun_out/c3b.log 2>&1
grep "^==" gpurun_out/c3a.log gpurun_out/c3b.log
python - <<'PY'
import json
for P in (4,0,2,8):
    try:
        d=json.load(open(f"gpurun_out/c3_trait_{P}.json")); t=d["trait_shaped"]
        print("parts",P,"commit+open",round(t["ms_per_commit_open"],2),"commit",round(t["commit_ms"],2),"open",round(t["open_ms"],2),"cache",round(t.get("with_shim_polynomial_cache_ms") or 0,2),t["parity_ok"], "| step", round(d["ms_per_step"],2), d["parity"]["commit_ok"], d["parity"]["open_ok"])
    except Exception as e: print(P, "failed", e)
PY
