#!/bin/bash
# One gpurun call: GPU tests, the benchmark lines of every configuration, then the rocprofv3 profiles.
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputest.log 2>&1 || { tail -40 gpurun_out/r2_gputest.log; exit 1; }
tail -2 gpurun_out/r2_gputest.log
for cfg in ${BENCH_CFGS:-pass cfg2 cfg3 cfg4 cfg5}; do
  steps=50; [ $cfg = cfg5 ] && steps=10
  python bench.py --config $cfg --steps $steps > gpurun_out/r2_bench_$cfg.json 2> gpurun_out/r2_bench_$cfg.err || { tail -20 gpurun_out/r2_bench_$cfg.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_bench_$cfg.json"))
cb=d.get("cpu_baseline") or {}; ca=d.get("cpu_baseline_all_cores") or {}
print("$cfg", round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "| cpu 1t", round(cb.get("value",0),3), "all", round(ca.get("value",0),3), ca.get("cores"), "|", d["per_kernel_us"])
PY
done
[ -n "$SKIP_PROF" ] || bash tools/collect_profiles.sh
