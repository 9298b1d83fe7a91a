# round 2, run J: new IPA verifier tests, RCCL path of bench.py with one rank, 3-waves-per-SIMD accumulate variant
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_ipa_gpu.py -q -x > gpurun_out/j_pytest_ipa.log 2>&1; tail -3 gpurun_out/j_pytest_ipa.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
PC_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --no-cpu-baseline --secondary-log-degree 0 --no-h2d > gpurun_out/j_dist_kzg.json 2> gpurun_out/j_dist_kzg.err; tail -2 gpurun_out/j_dist_kzg.err
PC_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --workload batch --steps 3 > gpurun_out/j_dist_batch.json 2> gpurun_out/j_dist_batch.err; tail -2 gpurun_out/j_dist_batch.err
PC_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --workload ntt --steps 20 > gpurun_out/j_dist_ntt.json 2> gpurun_out/j_dist_ntt.err; tail -2 gpurun_out/j_dist_ntt.err
unset MASTER_ADDR MASTER_PORT RANK LOCAL_RANK WORLD_SIZE
B="python bench.py --no-cpu-baseline"
timeout -k 10 600 $B > gpurun_out/j_base.json 2>/dev/null
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_w3.so timeout -k 10 600 $B > gpurun_out/j_w3.json 2>/dev/null
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_w3.so PC_HIP_TBL_LANES=196608 timeout -k 10 600 $B --secondary-log-degree 0 > gpurun_out/j_w3_l192k.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/j_*.json")):
    try:
        d = json.load(open(f))
        s = d.get("secondary") or {}
        print(f, round(d["ms_per_step"], 2), d.get("blocking_msm_ms"), {k: round(v, 2) for k, v in (d.get("msm_phase_ms") or {}).items()},
              "| 2^20", s.get("ms_per_step"), s.get("blocking_msm_ms"), (d.get("roofline") or {}).get("arithmetic"))
    except Exception as e: print(f, "failed", e)
PY
