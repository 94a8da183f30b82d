set -e
mkdir -p gpurun_out/r4a
python bench.py --config pass --steps 50 --no-cpu-baseline --no-boundary > gpurun_out/r4a/bench_pass.json 2> gpurun_out/r4a/bench_pass.err
python - <<PY
import json
d=json.load(open("gpurun_out/r4a/bench_pass.json"))
print("pass", round(d["value"],1), "ms", round(d["ms_per_step"],4), d.get("per_kernel_us"))
PY
RS_STAMPS=1 python tools/k78_stamps.py > gpurun_out/r4a/k78.txt 2>&1
RS_STAMPS=1 python tools/k5_stamps.py > gpurun_out/r4a/k5.txt 2>&1
cat gpurun_out/r4a/k78.txt gpurun_out/r4a/k5.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4a/raw -- python3 bench.py --config pass --steps 20 --warmup 2 --no-cpu-baseline --no-boundary > gpurun_out/r4a/prof.json 2> gpurun_out/r4a/prof.err
python3 tools/pmc_summarize.py stats gpurun_out/r4a/raw gpurun_out/r4a/pass_kernel_stats.csv
rm -rf gpurun_out/r4a/raw
cat gpurun_out/r4a/pass_kernel_stats.csv
