mkdir -p gpurun_out
make -s -C oracle
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for inf in 0 2 4; do
timeout -k 10 600 python bench.py --inflight $inf --no-cpu-baseline > gpurun_out/bench_r01_i_inf$inf.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/bench_r01_i_inf$inf.json'));print('inflight',$inf,d['value'],d['ms_per_step'],d['msm_phase_ms'])"
done
PC_HIP_SPLIT_CUS=0 timeout -k 10 600 python bench.py --inflight 2 --no-cpu-baseline 2>/dev/null | cut -c100-260
timeout -k 10 600 python bench.py --workload batch --steps 2 --warmup 1 2>/dev/null | cut -c100-330
PC_HIP_SPLIT_CUS=0 timeout -k 10 600 python bench.py --workload batch --steps 2 --warmup 1 2>/dev/null | cut -c100-330
