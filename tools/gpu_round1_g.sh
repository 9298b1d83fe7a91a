set -x
mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
PC_HIP_SORT=atomic timeout -k 10 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q 2>&1 | tail -2
for inf in 0 2; do
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --inflight $inf --no-cpu-baseline > gpurun_out/bench_r01_e_inf$inf.json 2> gpurun_out/bench_r01_e.err; tail -2 gpurun_out/bench_r01_e.err; python -c "
import json;d=json.load(open('gpurun_out/bench_r01_e_inf$inf.json'));print('inflight',$inf,d['value'],d['ms_per_step'],d['msm_phase_ms'])"
done
timeout -k 10 900 python bench.py --log-degree 24 --steps 3 --warmup 1 --no-cpu-baseline --window-bits 20 > gpurun_out/bench_r01_2p24_c20b.json 2> gpurun_out/bench_r01_2p24.err; tail -3 gpurun_out/bench_r01_2p24.err; python -c "
import json;d=json.load(open('gpurun_out/bench_r01_2p24_c20b.json'));print('2^24 c20',d['value'],d['ms_per_step'],d['msm_phase_ms'])"
