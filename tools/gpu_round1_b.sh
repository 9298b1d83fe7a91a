set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r01_a.json 2> gpurun_out/bench_r01_a.err; tail -3 gpurun_out/bench_r01_a.err; cat gpurun_out/bench_r01_a.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_a -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_a.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_a | head -20
