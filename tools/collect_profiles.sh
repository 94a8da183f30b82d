#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats + FETCH/WRITE counter passes of bench.py per configuration.
# Output: gpurun_out/prof_r2/{<cfg>_kernel_stats.csv, pmc_<cfg>.csv, bench_<cfg>.json}; copy into profiles/ to commit.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r2
mkdir -p $OUT
for cfg in ${CFGS:-pass cfg4 cfg5}; do
  steps=20; [ $cfg = cfg5 ] && steps=5
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_stats_$cfg -- python3 bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-boundary > $OUT/bench_prof_$cfg.json 2> $OUT/stats_$cfg.err
  python3 tools/pmc_summarize.py stats $OUT/raw_stats_$cfg $OUT/${cfg}_kernel_stats.csv
  echo "stats $cfg done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/raw_fetch_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-boundary > /dev/null 2> $OUT/fetch_$cfg.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/raw_write_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-boundary > /dev/null 2> $OUT/write_$cfg.err
  python3 tools/pmc_summarize.py pmc $OUT/raw_fetch_$cfg $OUT/raw_write_$cfg $OUT/pmc_$cfg.csv
  echo "pmc $cfg done"
  rm -rf $OUT/raw_stats_$cfg $OUT/raw_fetch_$cfg $OUT/raw_write_$cfg
done
