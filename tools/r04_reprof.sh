set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 900 python bench.py > gpurun_out/d_bench_n1.json 2> gpurun_out/d_bench.err; tail -2 gpurun_out/d_bench.err
cd /tmp && export TMPDIR=/tmp
B24="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-trait --inflight 0 --secondary-log-degree 0 --workloads none"
B20="python $R/bench.py --log-degree 20 --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-trait --inflight 0 --secondary-log-degree 0 --workloads none"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
prof() { name=$1; shift; timeout -k 10 600 rocprofv3 "$@" > $R/gpurun_out/$name.log 2>&1; }
for w in 24:"$B24" 20:"$B20"; do
  k=${w%%:*}; cmd=${w#*:}
  rm -rf $R/gpurun_out/d_prof$k $R/gpurun_out/d_fetch$k $R/gpurun_out/d_write$k $R/gpurun_out/d_sq$k
  prof d_prof$k --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_prof$k -o bench -- $cmd
  prof d_fetch$k --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/d_fetch$k -o bench -- $cmd
  prof d_write$k --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/d_write$k -o bench -- $cmd
  prof d_sq$k --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/d_sq$k -o bench -- $cmd
done
cd $R
find gpurun_out -name "*.csv" -size +30M -delete 2>/dev/null
head -3 gpurun_out/d_prof24/bench_kernel_stats.csv | cut -c1-170
