# target lanes of the accumulate grid for the 8-limb kernels (2 / 3 / 4 waves per SIMD), default and 128-VGPR builds
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
for v in default w4; do
  lib=$R/poly_commit_amd/libpc_hip.so; [ $v != default ] && lib=$R/poly_commit_amd/libpc_hip_$v.so
  for L in 131072 196608 262144 393216; do
    PC_HIP_TBL_LANES=$L PC_HIP_LIB=$lib timeout -k 10 400 python tools/n8_probe.py > gpurun_out/p2_probe_${v}_$L.json 2> gpurun_out/p2_probe_${v}_$L.err || tail -5 gpurun_out/p2_probe_${v}_$L.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/p2_probe_*.json")):
    try:
        d=json.load(open(f))
        print(f.split("p2_probe_")[1], "bn254 single", round(d['bn254_single']['blocking_ms'],3), d['bn254_single']['phases_ms'][3], round(d['bn254_single']['accumulate_madd_per_s']/1e9,2), "| batch", round(d['bn254_batch']['ms_per_step'],2), "| pallas", round(d['pallas_single']['blocking_ms'],3), d['pallas_single']['phases_ms'][3], round(d['pallas_single']['accumulate_madd_per_s']/1e9,2), d['bn254_single']['parity_ok'], d['bn254_batch']['parity_ok'], d['pallas_single']['parity_ok'])
    except Exception as e: print(f, "failed", e)
PY
