mkdir -p gpurun_out/r3e
timeout 600 python -m pytest tests/test_group_gpu.py tests/test_bench_gpu.py::test_group_mode_one_process tests/test_kzg_gpu.py -x -q 2>&1 | tail -3
for m in host device; do
  PC_HIP_GROUP_TRACE=1 timeout 300 python bench.py --mode group --gpus 1 --steps 12 --group-coeffs $m > gpurun_out/r3e/group_$m.json 2> gpurun_out/r3e/group_$m.err
  echo $m; grep "group job" gpurun_out/r3e/group_$m.err | tail -5; python -c "import json;d=json.load(open('gpurun_out/r3e/group_$m.json'));print(d['ms_per_step'])"
done
timeout 300 python bench.py --workloads none --steps 12 --secondary-log-degree 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('rank path',d['ms_per_step'])"
