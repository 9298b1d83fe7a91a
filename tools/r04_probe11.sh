set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
for cfg in 0:524288 64:1048576 128:1048576 256:1048576 128:2097152 192:524288; do
  C=${cfg%%:*}; L=${cfg#*:}
  PC_HIP_TBL_CHUNK=$C PC_HIP_TBL_MAX_LANES=$L timeout -k 10 400 python tools/n8_probe.py > gpurun_out/p11_probe_${C}_$L.json 2> gpurun_out/p11_probe_${C}_$L.err || tail -3 gpurun_out/p11_probe_${C}_$L.err
  PC_HIP_TBL_CHUNK=$C PC_HIP_TBL_MAX_LANES=$L timeout -k 10 400 python bench.py --steps 6 --no-cpu-baseline --workloads none --no-h2d > gpurun_out/p11_bench_${C}_$L.json 2>/dev/null
done
python - <<'PY'
import json
for cfg in ("0:524288","64:1048576","128:1048576","256:1048576","128:2097152","192:524288"):
    C,L=cfg.split(":")
    try:
        d=json.load(open(f"gpurun_out/p11_probe_{C}_{L}.json"))
        print(cfg, "bn254 single", round(d['bn254_single']['blocking_ms'],3), d['bn254_single']['phases_ms'][3], "| batch", round(d['bn254_batch']['ms_per_step'],2), "| pallas", round(d['pallas_single']['blocking_ms'],3), d['pallas_single']['phases_ms'][3], d['bn254_single']['parity_ok'], d['bn254_batch']['parity_ok'], d['pallas_single']['parity_ok'])
        d=json.load(open(f"gpurun_out/p11_bench_{C}_{L}.json")); s=d["secondary"]; t=d["trait_shaped"]
        print("    bls 2^24 step", round(d["ms_per_step"],2), "blocking", round(d["blocking_msm_ms"],2), "acc", round(d["msm_phase_ms"]["accumulate"],2), "seg", round(d["msm_phase_ms"]["seg_reduce"],2), "trait", round(t["ms_per_commit_open"],1), round(t["with_shim_polynomial_cache_ms"],1), "| 2^20 step", round(s["ms_per_step"],2), "blocking", round(s["blocking_msm_ms"],2), "acc", round(s["msm_phase_ms"]["accumulate"],2), "trait", round(s["trait_shaped"]["ms_per_commit_open"],2), d["parity"]["commit_ok"], s["parity"]["open_ok"])
    except Exception as e: print(cfg, "failed", e)
PY
