set -e
mkdir -p gpurun_out/bt
timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "bundle_adjust or ba_ or smoke or inertial or round4 or resident" > gpurun_out/bt/tests.log 2>&1 || { tail -60 gpurun_out/bt/tests.log; exit 1; }
tail -2 gpurun_out/bt/tests.log
timeout -k 10 120 python tools/ab_time.py
timeout -k 10 120 python tools/ab_time.py ba_speculative_sets=3
timeout -k 10 120 python tools/ab_time.py ba_speculative_sets=5
