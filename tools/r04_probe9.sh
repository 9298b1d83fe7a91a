set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
for v in default hot; do
  lib=$R/poly_commit_amd/libpc_hip.so; [ $v != default ] && lib=$R/poly_commit_amd/libpc_hip_$v.so
  PC_HIP_LIB=$lib timeout -k 10 400 python tools/n8_probe.py > gpurun_out/p9_probe_$v.json 2> gpurun_out/p9_probe_$v.err || tail -3 gpurun_out/p9_probe_$v.err
  PC_HIP_LIB=$lib timeout -k 10 400 python bench.py --steps 4 --no-cpu-baseline --workloads none --no-h2d --no-trait --secondary-log-degree 0 > gpurun_out/p9_bench_$v.json 2>/dev/null
done
python - <<'PY'
import json
for v in ("default","hot"):
    d=json.load(open(f"gpurun_out/p9_probe_{v}.json"))
    print(v, "bn254 single", d['bn254_single']['phases_ms'][3], round(d['bn254_single']['accumulate_madd_per_s']/1e9,2), "| batch", round(d['bn254_batch']['ms_per_step'],2), "| pallas", d['pallas_single']['phases_ms'][3], round(d['pallas_single']['accumulate_madd_per_s']/1e9,2))
    d=json.load(open(f"gpurun_out/p9_bench_{v}.json"))
    print(v, "bls 2^24 step", round(d["ms_per_step"],2), "blocking", round(d["blocking_msm_ms"],2), {k: round(x, 2) for k, x in d["msm_phase_ms"].items()})
PY
