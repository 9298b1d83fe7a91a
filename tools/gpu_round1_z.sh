make -s -C oracle
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c17 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 0 --window-bits 17 > $R/gpurun_out/prof_c17.log 2>&1
