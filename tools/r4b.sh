set -e
mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -x -q -k "bundle_adjust or ba_ or smoke or inertial or refine" > gpurun_out/r4b/tests.log 2>&1 || { tail -40 gpurun_out/r4b/tests.log; exit 1; }
tail -3 gpurun_out/r4b/tests.log
RS_LIB=librsgpu_k7v1.so python tools/ab_time.py
python tools/ab_time.py
RS_LIB=librsgpu_k7v1.so python tools/ab_time.py ba_fuse_mode=1
python tools/ab_time.py ba_fuse_mode=1
