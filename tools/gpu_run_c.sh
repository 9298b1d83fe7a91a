# GPU call C of round 2: suite after the scratch fixes / edge merge / serialized SRS; same-box A/B of the squaring; NTT variants.
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 120 tools/microbench_old > gpurun_out/c_microbench_old.txt 2>&1; timeout -k 10 120 tools/microbench > gpurun_out/c_microbench.txt 2>&1
grep -h "fmul bls\|fsqr bls\|madd" gpurun_out/c_microbench_old.txt gpurun_out/c_microbench.txt
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/c_pytest.log 2>&1; tail -12 gpurun_out/c_pytest.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 600 python bench.py --workload ntt > gpurun_out/c_ntt_default.json 2>/dev/null
PC_HIP_NTT_GROUP=0 timeout -k 10 600 python bench.py --workload ntt > gpurun_out/c_ntt_group0.json 2>/dev/null
PC_HIP_NTT_GROUP=64 timeout -k 10 600 python bench.py --workload ntt > gpurun_out/c_ntt_group64.json 2>/dev/null
PC_HIP_LIB=$R/poly-commit_amd/libpc_hip_ntt512.so timeout -k 10 600 python bench.py --workload ntt > gpurun_out/c_ntt_512.json 2>/dev/null
PC_HIP_LIB=$R/poly-commit_amd/libpc_hip_ntt512.so PC_HIP_NTT_GROUP=0 timeout -k 10 600 python bench.py --workload ntt > gpurun_out/c_ntt_512_group0.json 2>/dev/null
for f in gpurun_out/c_ntt_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['ntt_phase_ms'], d['column_hash_blake2s_ms'], d['merkle_tree_sha256_ms'])"; done
timeout -k 10 900 python bench.py > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err; tail -3 gpurun_out/c_bench.err
timeout -k 10 300 python bench.py --log-degree 20 --secondary-log-degree 0 --inflight 0 --no-h2d --no-cpu-baseline > gpurun_out/c_bench20_blocking.json 2>/dev/null
timeout -k 10 300 python tools/ipa_timing.py 22 2>/dev/null | tail -1 > gpurun_out/c_ipa_2p22.json; cat gpurun_out/c_ipa_2p22.json
cd /tmp && export TMPDIR=/tmp
B20="python $R/bench.py --log-degree 20 --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --inflight 0 --secondary-log-degree 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c_prof20 -o bench -- $B20 > $R/gpurun_out/c_prof20.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c_profntt -o bench -- python $R/bench.py --workload ntt --steps 3 --warmup 1 > $R/gpurun_out/c_profntt.log 2>&1
cd $R
python -c "
import json
d=json.load(open('gpurun_out/c_bench.json')); s=d['secondary']
print('2^24', d['ms_per_step'], d['blocking_msm_ms'], d['msm_phase_ms'])
print('2^20', s['ms_per_step'], s['blocking_msm_ms'], s['msm_phase_ms'])
print(json.load(open('gpurun_out/c_bench20_blocking.json'))['ms_per_step'])
"
