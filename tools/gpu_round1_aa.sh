make -s -C oracle
timeout -k 5 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for extra in "" "--log-degree 22 --steps 5" "--log-degree 18 --steps 20" "--log-degree 24 --steps 3 --warmup 1"; do
timeout -k 5 120 python bench.py --inflight 0 --no-cpu-baseline $extra > gpurun_out/tmp.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tmp.json'));p=d['msm_phase_ms'];print('[$extra]',round(d['ms_per_step'],2),round(d['value']/1e6,1),{k:round(v,2) for k,v in p.items()})"
done
timeout -k 5 120 python tools/ipa_timing.py 22 2>/dev/null | tail -1
