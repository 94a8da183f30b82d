# A/B of a variant library on the banded (cfg 5) path: BA tests on the variant, then both timed. usage: bash tools/ab_variant5.sh librsgpu_<name>.so
set -e
V=$1
mkdir -p gpurun_out/ab
RS_LIB=$V timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "bundle_adjust or ba_ or smoke or inertial or round4 or band or big or shard" > gpurun_out/ab/tests.log 2>&1 || { tail -60 gpurun_out/ab/tests.log; exit 1; }
tail -2 gpurun_out/ab/tests.log
RS_LIB=$V timeout -k 10 120 python tools/ab_time.py cfg5
timeout -k 10 120 python tools/ab_time.py cfg5
RS_LIB=$V timeout -k 10 120 python tools/ab_time.py cfg5
