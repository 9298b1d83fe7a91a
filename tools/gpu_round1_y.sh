make -s -C oracle
for extra in "" "--chunk 48" "--chunk 96" "--chunk 128" "--window-bits 15" "--window-bits 17" "--window-bits 18" ""; do
timeout -k 5 90 python bench.py --inflight 0 --no-cpu-baseline --steps 10 $extra > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));p=d['msm_phase_ms'];print('[$extra]',round(d['ms_per_step'],2),{k:round(v,2) for k,v in p.items()})"
done
