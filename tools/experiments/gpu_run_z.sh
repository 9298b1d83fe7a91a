# round 2, run Z: hipGraph replay of repeated MSM calls
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 1200 python -m pytest tests/test_msm_gpu.py tests/test_kzg_gpu.py tests/test_ipa_gpu.py tests/test_hyrax_gpu.py tests/test_group_gpu.py -q -x > gpurun_out/z_pytest.log 2>&1; tail -4 gpurun_out/z_pytest.log
timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/z_ipa_graphs.json
PC_HIP_GRAPHS=0 timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/z_ipa_nographs.json
B20="python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0"
timeout -k 10 600 $B20 > gpurun_out/z_2p20_graphs.json 2>/dev/null
PC_HIP_GRAPHS=0 timeout -k 10 600 $B20 > gpurun_out/z_2p20_nographs.json 2>/dev/null
timeout -k 10 600 $B20 --inflight 0 > gpurun_out/z_2p20_blocking_graphs.json 2>/dev/null
PC_HIP_GRAPHS=0 timeout -k 10 600 $B20 --inflight 0 > gpurun_out/z_2p20_blocking_nographs.json 2>/dev/null
timeout -k 10 600 python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0 > gpurun_out/z_2p24_graphs.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/z_*.json")):
    try:
        d = json.load(open(f))
        if "ms_per_step" in d: print(f, round(d["ms_per_step"], 3), d["steps"], round(d.get("blocking_msm_ms"),3))
        else: print(f, d.get("commit_ms"), d.get("open_rounds_ms"), d.get("open_breakdown_ms"))
    except Exception as e: print(f, "failed", e)
PY
