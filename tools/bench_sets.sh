set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r2_test2.log 2>&1 || { tail -40 gpurun_out/r2_test2.log; exit 1; }
tail -3 gpurun_out/r2_test2.log
for ns in 1 2 3; do RS_BA_SETS=$ns python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_ns$ns.json 2> gpurun_out/r2_bench_ns$ns.err || { tail -20 gpurun_out/r2_bench_ns$ns.err; exit 1; }; python - <<PY
import json
d=json.load(open("gpurun_out/r2_bench_ns$ns.json"))
print("ns=$ns", round(d["value"],1), "passes/s", d["ms_per_step"], d["per_kernel_us"], d["per_kernel_launches_per_pass"], d["ba_summary"])
PY
done
python tools/ba_time.py
