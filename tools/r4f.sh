set -e
mkdir -p gpurun_out/r4f
timeout -k 10 800 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "round4 or shard or banded or batch" > gpurun_out/r4f/tests.log 2>&1 || { tail -60 gpurun_out/r4f/tests.log; exit 1; }
tail -3 gpurun_out/r4f/tests.log
