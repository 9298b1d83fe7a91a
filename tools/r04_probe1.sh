# Round-4 probe of the 8-limb curves (one gpurun call): memory-free ceilings, three library variants, rocprofv3 kernel stats and
# SQ counters of BASELINE configs[2] (BN254 batch) and [3] (Pallas IPA).
set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 240 tools/microbench > gpurun_out/p1_microbench.txt 2>&1
tail -14 gpurun_out/p1_microbench.txt
for v in default nolazy8 w4; do
  lib=$R/poly_commit_amd/libpc_hip.so; [ $v != default ] && lib=$R/poly_commit_amd/libpc_hip_$v.so
  PC_HIP_LIB=$lib timeout -k 10 400 python tools/n8_probe.py > gpurun_out/p1_probe_$v.json 2> gpurun_out/p1_probe_$v.err || tail -5 gpurun_out/p1_probe_$v.err
  cat gpurun_out/p1_probe_$v.json
done
cd /tmp && export TMPDIR=/tmp
BATCH="python $R/bench.py --workload batch --steps 2 --warmup 1"
IPA="python $R/tools/ipa_timing.py 22"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
prof() { name=$1; shift; timeout -k 10 500 rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1; }
prof p1_prof_batch --kernel-trace --stats --output-format csv -d $R/gpurun_out/p1_prof_batch -o bench -- $BATCH
PC_IPA_REPS=2 prof p1_prof_ipa --kernel-trace --stats --output-format csv -d $R/gpurun_out/p1_prof_ipa -o bench -- $IPA
prof p1_sq_batch --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/p1_sq_batch -o bench -- $BATCH
prof p1_fetch_batch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p1_fetch_batch -o bench -- $BATCH
prof p1_write_batch --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p1_write_batch -o bench -- $BATCH
cd $R
find gpurun_out -name "*.csv" -size +30M -delete 2>/dev/null
for d in p1_prof_batch p1_prof_ipa; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -14 "$f" | cut -c1-200; done
