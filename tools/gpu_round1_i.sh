mkdir -p gpurun_out
make -s -C oracle
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_2p24 -o bench -- python $R/bench.py --log-degree 24 --steps 2 --warmup 1 --no-cpu-baseline --inflight 0 > $R/gpurun_out/prof_2p24.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --inflight 0 > $R/gpurun_out/prof_c.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ntt -o bench -- python $R/bench.py --workload ntt --steps 3 --warmup 1 > $R/gpurun_out/prof_ntt.log 2>&1
cd $R
tail -1 gpurun_out/prof_2p24.log | cut -c1-600
