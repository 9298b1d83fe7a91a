cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
run() { echo "== $1"; env $1 timeout -k 10 300 python tools/n8_probe.py 20 22 16 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('bn254 single', round(d['bn254_single']['blocking_ms'],3), d['bn254_single']['phases_ms'][3], 'batch16', round(d['bn254_batch']['ms_per_step'],2), 'pallas', round(d['pallas_single']['blocking_ms'],3), d['pallas_single']['phases_ms'][3], d['bn254_single']['parity_ok'], d['pallas_single']['parity_ok'])"; }
run "A=1"
run "PC_HIP_TBL_LANES=196608 PC_HIP_TBL_MAX_LANES=196608"
run "PC_HIP_TBL_LANES=196608 PC_HIP_TBL_MAX_LANES=393216"
run "PC_HIP_TBL_LANES=196608 PC_HIP_TBL_MAX_LANES=786432"
run "PC_HIP_TBL_LANES=131072 PC_HIP_TBL_MAX_LANES=131072"
run "PC_HIP_TBL_LANES=131072 PC_HIP_TBL_MAX_LANES=262144"
run "PC_HIP_TBL_LANES=131072 PC_HIP_TBL_MAX_LANES=1048576"
