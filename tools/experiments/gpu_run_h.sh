set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
if ! timeout -k 10 300 python -m pytest tests/test_msm_gpu.py -m gpu -q -x > gpurun_out/h_sanity.log 2>&1; then tail -30 gpurun_out/h_sanity.log; exit 1; fi
tail -2 gpurun_out/h_sanity.log
PC_HIP_LIB=$R/poly-commit_amd/libpc_hip_fine512.so timeout -k 10 300 python -m pytest tests/test_msm_gpu.py -m gpu -q -x 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d > gpurun_out/h_$tag.json 2>/dev/null; env "$@" timeout -k 10 300 python bench.py --workload batch --steps 5 > gpurun_out/h_batch_$tag.json 2>/dev/null; }
run default PC_X=0
run fine512 PC_HIP_LIB=$R/poly-commit_amd/libpc_hip_fine512.so
for k in 17 15 13 11; do timeout -k 10 300 python tools/ipa_timing.py 22 $k 2>/dev/null | tail -1 > gpurun_out/h_ipa_fkb$k.json; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/h_prof24 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --inflight 0 --secondary-log-degree 0 > $R/gpurun_out/h_prof24.log 2>&1
cd $R
python - <<'PY'
import json, csv
for t in ("default", "fine512"):
    d = json.load(open(f"gpurun_out/h_{t}.json")); s = d["secondary"]; b = json.load(open(f"gpurun_out/h_batch_{t}.json"))
    print(t, "2^24", round(d["ms_per_step"], 2), round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()},
          "| 2^20", round(s["ms_per_step"], 2), round(s["blocking_msm_ms"], 2), {k: round(v, 3) for k, v in s["msm_phase_ms"].items()}, "| batch", round(b["ms_per_step"], 1))
for k in (17, 15, 13, 11):
    d = json.loads(open(f"gpurun_out/h_ipa_fkb{k}.json").read()); print(k, round(d["open_rounds_ms"], 1), d["open_breakdown_ms"], d["per_round_ms"])
for r in csv.DictReader(open("gpurun_out/h_prof24/bench_kernel_stats.csv")):
    if float(r["AverageNs"]) > 1e5: print(f"  {r['Name'][:60]:60s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:8.3f} ms")
PY
