# round 2, run T: which kernels should raise their wave priority (all / all but sort / all but k_run / cooperative reductions only / none)
set -x
mkdir -p gpurun_out
make -s -C oracle
B20="python bench.py --no-cpu-baseline --no-h2d --log-degree 20 --secondary-log-degree 0"
B24="python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0"
for v in all nosort norun cooponly noprio; do
  L=$PWD/poly-commit_amd/libpc_hip_$v.so; [ $v = all ] && L=$PWD/poly-commit_amd/libpc_hip.so
  PC_HIP_LIB=$L timeout -k 10 600 $B20 > gpurun_out/t_2p20_$v.json 2>/dev/null
  PC_HIP_LIB=$L timeout -k 10 600 $B24 > gpurun_out/t_2p24_$v.json 2>/dev/null
  PC_HIP_LIB=$L timeout -k 10 600 python bench.py --workload batch > gpurun_out/t_batch_$v.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/t_*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 3), d["steps"], d.get("blocking_msm_ms"))
    except Exception as e: print(f, "failed", e)
PY
