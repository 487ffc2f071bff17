set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_exchange" > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
