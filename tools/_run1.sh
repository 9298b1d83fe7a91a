timeout 300 python tools/hyrax_timing.py 2>/dev/null | grep workload | cut -c1-260
timeout 600 python -m pytest tests/test_hyrax_gpu.py tests/test_msm_gpu.py -x -q 2>&1 | tail -2
