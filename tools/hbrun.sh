set -e
timeout -k 10 300 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_samplers.py tests/test_gpu_threads.py -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/time_grid.py 2 8 | grep "hb\|chain"
timeout -k 10 100 python tools/time_smake.py 10000
