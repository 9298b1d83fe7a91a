#!/bin/bash
# Copies the summaries of one tools/gpu_full_run.sh call (gpurun_out/d_*) into profiles/<round>_* (tracked).  usage: collect_profiles.sh r05
set -e
cd "$(dirname "$0")/.."
P=${1:-r06}
G=gpurun_out
cp $G/d_microbench.txt profiles/${P}_microbench.txt
for f in bench_n1 bench_n1_inflight0 bench_n1_notable bench_n1_glv bench_ntt bench_rccl_1rank bench_2ranks_dev0 bench_8ranks_dev0 group_host group_device; do [ -f $G/d_$f.json ] && cp $G/d_$f.json profiles/${P}_$f.json; done
[ -f $G/d_bench_n1_line.json ] && cp $G/d_bench_n1_line.json profiles/${P}_bench_n1_line.json      # the line the driver parses (<= 6 KB); ${P}_bench_n1.json is the full record it points at
cp $G/d_bench_batch.json profiles/${P}_bench_batch_bn254.json
cp $G/d_ipa_2p22.json profiles/${P}_ipa_pallas_2p22.json
cp $G/d_lincomb.json profiles/${P}_lincomb_bn254.json
[ -f $G/d_hyrax.jsonl ] && cp $G/d_hyrax.jsonl profiles/${P}_hyrax_bn254.jsonl
[ -f $G/d_msm_size_sweep.json ] && cp $G/d_msm_size_sweep.json profiles/${P}_msm_size_sweep.json
[ -f $G/d_host_parts.jsonl ] && cp $G/d_host_parts.jsonl profiles/${P}_host_parts.jsonl
[ -f $G/d_ligero_stream.txt ] && cp $G/d_ligero_stream.txt profiles/${P}_ligero_stream.txt
[ -f $G/d_ldsntt/bench_counter_collection.csv ] && python tools/lds_summary.py $G/d_ldsntt/bench_counter_collection.csv profiles/${P}_ntt_lds_conflicts.json
cp $G/d_prof24/bench_kernel_stats.csv profiles/${P}_bench_2p24_kernel_stats.csv
cp $G/d_prof20/bench_kernel_stats.csv profiles/${P}_bench_2p20_kernel_stats.csv
cp $G/d_profntt/bench_kernel_stats.csv profiles/${P}_ntt_kernel_stats.csv
cp $G/d_profbatch/bench_kernel_stats.csv profiles/${P}_batch_bn254_kernel_stats.csv
cp $G/d_profpallas/bench_kernel_stats.csv profiles/${P}_pallas_2p22_msm_kernel_stats.csv
[ -f $G/d_profipa/bench_kernel_stats.csv ] && cp $G/d_profipa/bench_kernel_stats.csv profiles/${P}_ipa_kernel_stats.csv
rm -f profiles/${P}_pmc_traffic.json
python tools/pmc_summary.py $G/d_fetch24/bench_counter_collection.csv $G/d_write24/bench_counter_collection.csv profiles/${P}_pmc_traffic.json "bls12_381:2^24:table"
python tools/pmc_summary.py $G/d_fetch20/bench_counter_collection.csv $G/d_write20/bench_counter_collection.csv profiles/${P}_pmc_traffic.json "bls12_381:2^20:table"
python tools/pmc_summary.py $G/d_fetchntt/bench_counter_collection.csv $G/d_writentt/bench_counter_collection.csv profiles/${P}_pmc_traffic.json "ntt:bls12_381:2^24"
python tools/pmc_summary.py $G/d_fetchbatch/bench_counter_collection.csv $G/d_writebatch/bench_counter_collection.csv profiles/${P}_pmc_traffic.json "bn254:batch64x2^20:table"
python tools/pmc_summary.py $G/d_fetchpallas/bench_counter_collection.csv $G/d_writepallas/bench_counter_collection.csv profiles/${P}_pmc_traffic.json "pallas:2^22:table"
python tools/sq_summary.py kzg_2p24=$G/d_sq24/bench_counter_collection.csv kzg_2p20=$G/d_sq20/bench_counter_collection.csv ligero_2p24=$G/d_sqntt/bench_counter_collection.csv batch_bn254=$G/d_sqbatch/bench_counter_collection.csv pallas_2p22=$G/d_sqpallas/bench_counter_collection.csv profiles/${P}_valu.json | grep "accumulate\|ntt_pass\|ColumnHash"
