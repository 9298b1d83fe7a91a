make -s -C oracle
for rep in 1 2; do
for lib in libpc_hip_prev.so libpc_hip.so; do
PC_HIP_LIB=$PWD/poly-commit_amd/$lib timeout -k 5 90 python bench.py --inflight 0 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));print('$lib',d['ms_per_step'],d['msm_phase_ms'])"
done; done
