make -s -C oracle
for cfg in "PC_HIP_SEG_TAIL=256 PC_HIP_T2=8" "PC_HIP_SEG_TAIL=1 PC_HIP_T2=8" "PC_HIP_SEG_TAIL=1 PC_HIP_T2=16" "PC_HIP_SEG_TAIL=1 PC_HIP_T2=32" "PC_HIP_SEG_TAIL=64 PC_HIP_T2=16"; do
env $cfg timeout -k 5 60 python bench.py --inflight 0 --no-cpu-baseline --steps 10 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));print('$cfg',d['ms_per_step'],d['msm_phase_ms']['seg_reduce'],d['msm_phase_ms']['accumulate'])"
done
