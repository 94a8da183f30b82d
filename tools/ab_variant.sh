# A/B of a variant library (RS_LIB=...) against the product: BA tests on the variant, then both timed. usage: bash tools/ab_variant.sh librsgpu_<name>.so
set -e
V=$1
mkdir -p gpurun_out/ab
RS_LIB=$V timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "bundle_adjust or ba_ or smoke or inertial or round4" > gpurun_out/ab/tests.log 2>&1 || { tail -40 gpurun_out/ab/tests.log; exit 1; }
tail -2 gpurun_out/ab/tests.log
RS_LIB=$V timeout -k 10 120 python tools/ab_time.py
timeout -k 10 120 python tools/ab_time.py
RS_LIB=$V timeout -k 10 120 python tools/ab_time.py
timeout -k 10 120 python tools/ab_time.py
