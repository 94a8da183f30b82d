set -e
mkdir -p gpurun_out/r4d
python -m pytest tests -m gpu -x -q -k "bundle_adjust or ba_ or smoke or inertial" > gpurun_out/r4d/tests.log 2>&1 || { tail -40 gpurun_out/r4d/tests.log; exit 1; }
tail -2 gpurun_out/r4d/tests.log
python tools/ab_time.py
RS_LIB=librsgpu_k5seq.so python tools/ab_time.py
python tools/ab_time.py
RS_LIB=librsgpu_k5seq.so python tools/ab_time.py
