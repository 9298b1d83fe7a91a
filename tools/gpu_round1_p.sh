mkdir -p gpurun_out
make -s -C oracle
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_valu -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 0 > $R/gpurun_out/pmc_valu.log 2>&1
tail -2 $R/gpurun_out/pmc_valu.log | cut -c1-300
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 0 > $R/gpurun_out/pmc_fetch2.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 0 > $R/gpurun_out/pmc_write2.log 2>&1
