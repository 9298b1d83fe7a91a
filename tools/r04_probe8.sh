set -x
mkdir -p gpurun_out
make -s -C oracle
for G in 8 16 32 64; do
PC_HIP_BATCH_G=$G timeout -k 10 600 python bench.py --workload batch --steps 4 --warmup 2 > gpurun_out/p8_batch_G$G.json 2>gpurun_out/p8_batch_G$G.err || tail -3 gpurun_out/p8_batch_G$G.err
done
python - <<'PY'
import json
for G in (8,16,32,64):
    try:
        d=json.load(open(f"gpurun_out/p8_batch_G{G}.json"))
        r=d["roofline"]
        print("G",G, round(d["ms_per_step"],2), d["parity"]["all_commitments_closed_form_ok"], "passes", r["launches"], "kernel_ms", round(r["kernel_ms"],2), "arith", round(r["arithmetic"]["frac"],3), r["pass_phase_ms_sum"])
    except Exception as e: print(G, "failed", e)
PY
