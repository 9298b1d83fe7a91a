# round 2, run S: raised wave priority (s_setprio) in every kernel except the bucket accumulation
set -x
mkdir -p gpurun_out
make -s -C oracle
B20="python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0"
B24="python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0"
for rep in 1 2; do
  timeout -k 10 600 $B20 > gpurun_out/s_2p20_prio_$rep.json 2>/dev/null
  PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_noprio.so timeout -k 10 600 $B20 > gpurun_out/s_2p20_noprio_$rep.json 2>/dev/null
done
timeout -k 10 600 $B24 > gpurun_out/s_2p24_prio.json 2>/dev/null
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_noprio.so timeout -k 10 600 $B24 > gpurun_out/s_2p24_noprio.json 2>/dev/null
timeout -k 10 600 python bench.py --workload batch > gpurun_out/s_batch_prio.json 2>/dev/null
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_noprio.so timeout -k 10 600 python bench.py --workload batch > gpurun_out/s_batch_noprio.json 2>/dev/null
timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/s_ipa_prio.json
timeout -k 10 900 python -m pytest tests/test_msm_gpu.py tests/test_kzg_gpu.py -q -x 2>&1 | tail -2
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/s_*.json")):
    try:
        d = json.load(open(f))
        if "ms_per_step" in d: print(f, round(d["ms_per_step"], 3), d["steps"], d.get("blocking_msm_ms"), d.get("msm_phase_ms"))
        else: print(f, d.get("commit_ms"), d.get("open_rounds_ms"))
    except Exception as e: print(f, "failed", e)
PY
