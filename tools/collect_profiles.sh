#!/bin/bash
# Copies the summaries of one tools/gpu_full_run.sh call (gpurun_out/d_*) into profiles/r03_* (tracked).
set -e
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/d_microbench.txt profiles/r03_microbench.txt
for f in bench_n1 bench_n1_inflight0 bench_n1_notable bench_ntt bench_rccl_1rank bench_2ranks_dev0 group_host group_device; do [ -f $G/d_$f.json ] && cp $G/d_$f.json profiles/r03_$f.json; done
cp $G/d_bench_batch.json profiles/r03_bench_batch_bn254.json
cp $G/d_ipa_2p22.json profiles/r03_ipa_pallas_2p22.json
cp $G/d_lincomb.json profiles/r03_lincomb_bn254.json
[ -f $G/d_hyrax.jsonl ] && cp $G/d_hyrax.jsonl profiles/r03_hyrax_bn254.jsonl
[ -f $G/d_msm_size_sweep.json ] && cp $G/d_msm_size_sweep.json profiles/r03_msm_size_sweep.json
cp $G/d_prof24/bench_kernel_stats.csv profiles/r03_bench_2p24_kernel_stats.csv
cp $G/d_prof20/bench_kernel_stats.csv profiles/r03_bench_2p20_kernel_stats.csv
cp $G/d_profntt/bench_kernel_stats.csv profiles/r03_ntt_kernel_stats.csv
rm -f profiles/r03_pmc_traffic.json
python tools/pmc_summary.py $G/d_fetch24/bench_counter_collection.csv $G/d_write24/bench_counter_collection.csv profiles/r03_pmc_traffic.json "bls12_381:2^24:table"
python tools/pmc_summary.py $G/d_fetch20/bench_counter_collection.csv $G/d_write20/bench_counter_collection.csv profiles/r03_pmc_traffic.json "bls12_381:2^20:table"
python tools/pmc_summary.py $G/d_fetchntt/bench_counter_collection.csv $G/d_writentt/bench_counter_collection.csv profiles/r03_pmc_traffic.json "ntt:bls12_381:2^24"
python tools/sq_summary.py kzg_2p24=$G/d_sq24/bench_counter_collection.csv kzg_2p20=$G/d_sq20/bench_counter_collection.csv ligero_2p24=$G/d_sqntt/bench_counter_collection.csv profiles/r03_valu.json | grep "accumulate\|ntt_pass\|ColumnHash"
