set -e
mkdir -p gpurun_out/r4h
for cfg in cfg3 cfg5 cfg4 cfg2; do
  steps=50; [ $cfg = cfg5 ] && steps=10
  timeout -k 10 500 python bench.py --config $cfg --steps $steps --no-cpu-baseline > gpurun_out/r4h/bench_$cfg.json 2> gpurun_out/r4h/bench_$cfg.err || { tail -20 gpurun_out/r4h/bench_$cfg.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r4h/bench_$cfg.json"))
print("$cfg", round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "|", d["per_kernel_us"])
for k in ("batch", "in_process_shards", "batched_windows"):
    if k in d: print("   ", k, json.dumps(d[k])[:1500])
PY
done
