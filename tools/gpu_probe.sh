#!/bin/bash
# ONE parameterised GPU probe (a single gpurun call), replacing the round-4 one-off scripts.  It runs a list of commands for every
# library variant given, optionally under rocprofv3, and prints a one-line summary of every JSON result.
#
#   gpurun --timeout 900 -- 'VARIANTS="default nttr8" RUNS="ntt batch" PROF="ntt" SQ="ntt" bash tools/gpu_probe.sh p1'
#
#   $1        tag: outputs go to gpurun_out/<tag>_*
#   VARIANTS  library builds to compare: "default" = poly_commit_amd/libpc_hip.so, any other name = libpc_hip_<name>.so (built here on
#             the CPU beforehand: PC_HIP_VARIANT=<name> PC_HIP_CXXFLAGS="-D..." python -m poly_commit_amd.build)
#   RUNS      timed runs per variant, any of:
#               kzg24 kzg20   bench.py KZG commit+open line (pipelined; blocking MSM and phases in the line)   [KZG_FLAGS extra flags]
#               ntt           bench.py --workload ntt (config 5)
#               batch         bench.py --workload batch (config 3)
#               ipa           tools/ipa_timing.py 22 (config 4)
#               n8            tools/n8_probe.py (BN254 2^20 + Pallas 2^22 blocking MSMs with phase brackets)
#               trait         tools/pcie_inclusive.py (blocking commit + open from pageable host memory)
#               micro         tools/microbench
#               tests         pytest -m gpu (TESTS="-k expr" narrows it)
#   PROF      runs (same names) to repeat under rocprofv3 --kernel-trace --stats for the FIRST variant; SQ: the same with the SQ counters;
#             PMC: FETCH_SIZE and WRITE_SIZE passes
#   LDSC      runs to repeat with the LDS bank-conflict counters
#   ENVS      extra "NAME=value" pairs exported for every run (tuning environment variables of the library)
set -x
TAG=${1:-probe}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
export PC_BENCH_FULL_LINE=1          # full records on stdout (the driver's run gets the short line + bench_detail.json)
make -s -C oracle
for kv in $ENVS; do export "$kv"; done
VARIANTS=${VARIANTS:-default}
cmd_of() {
  case $1 in
    kzg24) echo "python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --workloads none --no-h2d --no-trait --secondary-log-degree 0 $KZG_FLAGS" ;;
    kzg20) echo "python $R/bench.py --log-degree 20 --steps 20 --warmup 3 --no-cpu-baseline --workloads none --no-h2d --no-trait --secondary-log-degree 0 $KZG_FLAGS" ;;
    ntt)   echo "python $R/bench.py --workload ntt --steps ${NTT_STEPS:-20} --warmup 2 --no-cpu-baseline" ;;
    batch) echo "python $R/bench.py --workload batch --steps ${BATCH_STEPS:-5} --warmup 1 --no-cpu-baseline" ;;
    ipa)   echo "python $R/tools/ipa_timing.py 22" ;;
    n8)    echo "python $R/tools/n8_probe.py" ;;
    trait) echo "python $R/tools/pcie_inclusive.py" ;;
    micro) echo "$R/tools/microbench" ;;
    tests) echo "python -m pytest $R/tests -m gpu -q -x $TESTS" ;;
  esac
}
for v in $VARIANTS; do
  lib=$R/poly_commit_amd/libpc_hip.so; [ $v != default ] && lib=$R/poly_commit_amd/libpc_hip_$v.so
  for r in $RUNS; do
    PC_HIP_LIB=$lib timeout -k 10 ${RUN_TIMEOUT:-600} $(cmd_of $r) > gpurun_out/${TAG}_${r}_$v.out 2> gpurun_out/${TAG}_${r}_$v.err || tail -5 gpurun_out/${TAG}_${r}_$v.err
  done
done
v0=${VARIANTS%% *}
lib0=$R/poly_commit_amd/libpc_hip.so; [ $v0 != default ] && lib0=$R/poly_commit_amd/libpc_hip_$v0.so
cd /tmp && export TMPDIR=/tmp
SQC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
prof() { name=$1; shift; PC_HIP_LIB=$lib0 NTT_STEPS=3 BATCH_STEPS=2 PC_IPA_REPS=2 timeout -k 10 600 rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1; }
for r in $PROF; do NTT_STEPS=3 BATCH_STEPS=2 prof ${TAG}_prof_$r --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_$r -o bench -- $(NTT_STEPS=3 BATCH_STEPS=2 cmd_of $r); done
for r in $SQ; do prof ${TAG}_sq_$r --kernel-trace --pmc $SQC --output-format csv -d $R/gpurun_out/${TAG}_sq_$r -o bench -- $(NTT_STEPS=3 BATCH_STEPS=2 cmd_of $r); done
# LDS: bank-conflict cycles against all LDS-array cycles (MI355X_MICROARCH.md, LDS section)
for r in $LDSC; do prof ${TAG}_lds_$r --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/${TAG}_lds_$r -o bench -- $(NTT_STEPS=3 BATCH_STEPS=2 cmd_of $r); done
for r in $PMC; do
  prof ${TAG}_fetch_$r --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch_$r -o bench -- $(NTT_STEPS=3 BATCH_STEPS=2 cmd_of $r)
  prof ${TAG}_write_$r --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write_$r -o bench -- $(NTT_STEPS=3 BATCH_STEPS=2 cmd_of $r)
done
cd $R
find gpurun_out -name "*.csv" -size +30M -delete 2>/dev/null
python tools/probe_summary.py $TAG
