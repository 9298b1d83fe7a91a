timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do timeout 300 python bench.py --workloads none --steps 12 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2^24', round(d['ms_per_step'],3), round(d['blocking_msm_ms'],3), d['roofline']['kernel_ms'], d['roofline']['serial']['kernel_ms'], '| 2^20', d['secondary']['ms_per_step'], d['secondary']['roofline']['serial']['kernel_ms'], d['parity']['commit_ok'], d['parity']['open_ok'])"; done
