# final state check of round 2: whole GPU suite, smoke, default bench line, batch group-size sweep
set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > gpurun_out/f_pytest.log 2>&1; tail -3 gpurun_out/f_pytest.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err; tail -2 gpurun_out/f_bench.err
for g in 4 8 16; do
  PC_HIP_BATCH_G=$g timeout -k 10 600 python bench.py --workload batch > gpurun_out/f_batch_g$g.json 2>/dev/null
done
python - <<'PY'
import json, glob
d = json.load(open("gpurun_out/f_bench.json")); s = d["secondary"]
print("bench", round(d["ms_per_step"], 2), round(d["value"] / 1e8, 3), "| 2^20", round(s["ms_per_step"], 3), "| cpu", d["cpu_baseline"]["value"], "| arith", d["roofline"]["arithmetic"]["frac"])
for f in sorted(glob.glob("gpurun_out/f_batch_g*.json")):
    x = json.load(open(f)); print(f, round(x["ms_per_step"], 2))
PY
