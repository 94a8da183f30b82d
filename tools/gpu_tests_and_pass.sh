set -e
mkdir -p gpurun_out/tp
python -m pytest tests -m gpu -x -q > gpurun_out/tp/tests.log 2>&1 || { tail -40 gpurun_out/tp/tests.log; exit 1; }
tail -2 gpurun_out/tp/tests.log
python bench.py --config pass --steps 50 --no-cpu-baseline --no-boundary > gpurun_out/tp/bench_pass.json 2> gpurun_out/tp/bench_pass.err
python - <<PY
import json
d=json.load(open("gpurun_out/tp/bench_pass.json"))
print("pass", round(d["value"],1), "ms", round(d["ms_per_step"],4), d.get("per_kernel_us"))
PY
