set -e
mkdir -p gpurun_out/r4g
timeout -k 10 600 python bench.py --config pass --steps 50 > gpurun_out/r4g/bench_pass.json 2> gpurun_out/r4g/bench_pass.err || { tail -30 gpurun_out/r4g/bench_pass.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r4g/bench_pass.json"))
print("pass", round(d["value"],1), "ms", round(d["ms_per_step"],4))
print(json.dumps(d["pass_through_boundary_ms"], indent=1))
b=d["boundary"]
for k,v in b.items():
    if isinstance(v,dict): print(k, v.get("median_us"), {kk:vv for kk,vv in v.items() if kk not in ("median_us","p10_us","p90_us","note")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"] if d["cpu_baseline_all_cores"] else None)
PY
