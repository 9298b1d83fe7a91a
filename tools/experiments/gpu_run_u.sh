# round 2, run U: kernel trace of the pipelined 2^20 run after the priority change
set -x
mkdir -p gpurun_out
make -s -C oracle
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/u_trace20 -o bench -- python $R/bench.py --log-degree 20 --steps 12 --warmup 3 --no-cpu-baseline --no-h2d --secondary-log-degree 0 > $R/gpurun_out/u_trace20.log 2>&1
cd $R
find gpurun_out/u_trace20 -name "*.csv" -size +30M -delete 2>/dev/null
ls -la gpurun_out/u_trace20/
