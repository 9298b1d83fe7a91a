make -s -C oracle
for lib in libpc_hip.so libpc_hip_w3.so; do
PC_HIP_LIB=$PWD/poly-commit_amd/$lib timeout -k 10 600 python bench.py --inflight 0 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));print('$lib',d['value'],d['ms_per_step'],d['msm_phase_ms']['accumulate'])"
done
PC_HIP_LIB=$PWD/poly-commit_amd/libpc_hip_w3.so timeout -k 10 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q 2>&1 | tail -2
