# The measurement sequence behind profiles/rNN_*: one gpurun call, one box.  (The summaries are copied to profiles/ by
# tools/collect_profiles.sh afterwards.)
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export PC_BENCH_FULL_LINE=1          # the runs below record bench.py's FULL record; the driver's short line is taken once (d_bench_n1_line.json)
make -s -C oracle
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout -k 10 240 tools/microbench > gpurun_out/d_microbench.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then PC_BENCH_FULL_LINE= timeout -k 10 1800 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/d_pytest.log 2>&1; tail -8 gpurun_out/d_pytest.log; fi
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
PC_BENCH_FULL_LINE= PC_BENCH_DETAIL=$R/gpurun_out/d_bench_n1.json timeout -k 10 900 python bench.py > gpurun_out/d_bench_n1_line.json 2> gpurun_out/d_bench.err; tail -2 gpurun_out/d_bench.err; wc -c gpurun_out/d_bench_n1_line.json
timeout -k 10 600 python bench.py --inflight 0 --no-cpu-baseline --workloads none > gpurun_out/d_bench_n1_inflight0.json 2>/dev/null
timeout -k 10 600 python bench.py --precompute 0 --no-cpu-baseline --workloads none > gpurun_out/d_bench_n1_notable.json 2>/dev/null
timeout -k 10 600 python bench.py --glv-table 1 --no-cpu-baseline --workloads none > gpurun_out/d_bench_n1_glv.json 2>/dev/null
timeout -k 10 600 python bench.py --workload ntt > gpurun_out/d_bench_ntt.json 2>/dev/null
(export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1; PC_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --no-cpu-baseline --no-h2d --secondary-log-degree 0 --workloads none > gpurun_out/d_bench_rccl_1rank.json 2>/dev/null)
# the N > 1 path as the driver would start it (self-launched ranks), two and eight ranks on this one GPU (gloo: RCCL refuses ranks that share a device)
PC_BENCH_DEVICES=0,0 timeout -k 10 600 python bench.py --gpus 2 --log-degree 23 --no-cpu-baseline > gpurun_out/d_bench_2ranks_dev0.json 2>/dev/null
PC_BENCH_DEVICES=0,0,0,0,0,0,0,0 timeout -k 10 900 python bench.py --gpus 8 --log-degree 21 --no-cpu-baseline > gpurun_out/d_bench_8ranks_dev0.json 2>/dev/null
# one process driving the device through the group API (host coefficients / resident shards)
timeout -k 10 300 python bench.py --mode group --gpus 1 --steps 12 --group-coeffs host > gpurun_out/d_group_host.json 2>/dev/null
timeout -k 10 300 python bench.py --mode group --gpus 1 --steps 12 --group-coeffs device > gpurun_out/d_group_device.json 2>/dev/null
timeout -k 10 600 python bench.py --workload batch > gpurun_out/d_bench_batch.json 2>/dev/null
timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/d_ipa_2p22.json
timeout -k 10 200 python tools/lincomb_timing.py 2>/dev/null | tail -1 > gpurun_out/d_lincomb.json
timeout -k 10 300 python tools/hyrax_timing.py 2>/dev/null | grep workload > gpurun_out/d_hyrax.jsonl
PC_SWEEP_LOGS=8,10,12,14,16,18,20,22 timeout -k 10 300 python tools/msm_size_sweep.py 2>/dev/null | tail -1 > gpurun_out/d_msm_size_sweep.json
# blocking calls on host memory (the trait-shaped calls): ONE MSM in parts (default weights), four equal parts, and round 4's two half-size MSMs
rm -f gpurun_out/d_host_parts.jsonl
for P in "1,2,5,8" "4" "0"; do PC_HIP_HOST_PARTS=$P timeout -k 10 300 python tools/host_parts_probe.py 24 2>/dev/null | tail -1 >> gpurun_out/d_host_parts.jsonl; done
# pc_hip_ligero_commit host -> host against the slab size (0 = whole matrix)
timeout -k 10 300 python tools/ligero_stream_probe.py --cold 2>/dev/null | grep slab > gpurun_out/d_ligero_stream.txt

cd /tmp && export TMPDIR=/tmp
B24="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-trait --inflight 0 --secondary-log-degree 0 --workloads none"
B20="python $R/bench.py --log-degree 20 --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-trait --inflight 0 --secondary-log-degree 0 --workloads none"
NTT="python $R/bench.py --workload ntt --steps 3 --warmup 1 --no-trait --no-cpu-baseline"   # config 5 ALONE: no slab launches of the host-to-host leg in the per-kernel averages
BATCH="python $R/bench.py --workload batch --steps 2 --warmup 1"
PAL="python $R/tools/msm_one.py pallas 22 6"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
prof() { name=$1; shift; timeout -k 10 600 rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1; }
for w in 24:"$B24" 20:"$B20" ntt:"$NTT" batch:"$BATCH" pallas:"$PAL"; do
  k=${w%%:*}; cmd=${w#*:}
  prof d_prof$k --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_prof$k -o bench -- $cmd
  prof d_fetch$k --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/d_fetch$k -o bench -- $cmd
  prof d_write$k --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/d_write$k -o bench -- $cmd
  prof d_sq$k --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/d_sq$k -o bench -- $cmd
done
PC_IPA_REPS=2 prof d_profipa --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_profipa -o bench -- python $R/tools/ipa_timing.py 22
# LDS bank conflicts of the NTT passes (conflict cycles against all LDS-array cycles)
prof d_ldsntt --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/d_ldsntt -o bench -- $NTT
cd $R
find gpurun_out -name "*.csv" -size +30M -delete 2>/dev/null
python - <<'PY'
import json
for f in ("d_bench_n1", "d_bench_n1_inflight0", "d_bench_n1_notable", "d_bench_n1_glv", "d_bench_rccl_1rank", "d_bench_2ranks_dev0", "d_bench_8ranks_dev0"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); s = d.get("secondary")
        print(f, "primary", round(d["ms_per_step"], 2), round(d["blocking_msm_ms"], 2), {k: round(v, 2) for k, v in d["msm_phase_ms"].items()},
              "| 2^20", (round(s["ms_per_step"], 2), round(s["blocking_msm_ms"], 2)) if s else None, d["parity"]["commit_ok"], d["parity"]["open_ok"])
    except Exception as e:
        print(f, "failed", e)
for f in ("d_group_host", "d_group_device"):
    d = json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["ms_per_step"], 2), d["parity"])
for f in ("d_bench_ntt", "d_bench_batch"):
    d = json.load(open(f"gpurun_out/{f}.json")); print(f, d["ms_per_step"], d.get("ntt_phase_ms"), d.get("column_hash_blake2s_ms"))
print(open("gpurun_out/d_ipa_2p22.json").read()[:600])
print(open("gpurun_out/d_host_parts.jsonl").read()[:2500])
PY
