make -s -C oracle
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
P='import json,sys
for l in sys.stdin:
    d=json.loads(l); print(sys.argv[1], "ms/step",round(d["ms_per_step"],3), "%.3e"%d["value"], {k:round(v,3) for k,v in d["msm_phase_ms"].items()})'
timeout -k 10 300 python bench.py --no-cpu-baseline --precompute 1 2>/dev/null | python -c "$P" "inflight2 pre1"
timeout -k 10 300 python bench.py --no-cpu-baseline --precompute 0 2>/dev/null | python -c "$P" "inflight2 pre0 K0=4"
PC_HIP_K0=8 timeout -k 10 300 python bench.py --no-cpu-baseline --precompute 0 2>/dev/null | python -c "$P" "inflight2 pre0 K0=8"
PC_HIP_K0=8 timeout -k 10 300 python bench.py --no-cpu-baseline --precompute 0 --inflight 0 2>/dev/null | python -c "$P" "inflight0 pre0 K0=8"
timeout -k 10 300 python bench.py --no-cpu-baseline --precompute 0 --inflight 0 2>/dev/null | python -c "$P" "inflight0 pre0 K0=4"
timeout -k 10 300 python bench.py --no-cpu-baseline --precompute 1 --inflight 3 2>/dev/null | python -c "$P" "inflight3 pre1"
timeout -k 10 300 python bench.py --no-cpu-baseline --precompute 1 --log-degree 22 --steps 5 2>/dev/null | python -c "$P" "2^22 pre1"
timeout -k 10 300 python bench.py --no-cpu-baseline --precompute 0 --log-degree 22 --steps 5 2>/dev/null | python -c "$P" "2^22 pre0"
