#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats + FETCH / WRITE / MFMA counter passes of bench.py per configuration,
# and the plain bench lines.  Counters are collected in runs of their own with --kernel-trace only (no --stats, no other
# trace domain).  Output: gpurun_out/prof_r4/{<cfg>_kernel_stats.csv, pmc_<cfg>.csv, pmc_mfma_pass.csv, bench_<cfg>.json};
# copy into profiles/ (round4_*) to commit.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r4
mkdir -p $OUT
for cfg in ${BENCH_CFGS:-pass cfg2 cfg3 cfg4 cfg5}; do
  steps=50; [ $cfg = cfg5 ] && steps=10
  python3 bench.py --config $cfg --steps $steps > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  echo "bench $cfg done"
done
for cfg in ${CFGS:-pass cfg4 cfg5}; do
  steps=20; [ $cfg = cfg5 ] && steps=5
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_stats_$cfg -- python3 bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-boundary --no-shard-rehearsal > $OUT/bench_prof_$cfg.json 2> $OUT/stats_$cfg.err
  python3 tools/pmc_summarize.py stats $OUT/raw_stats_$cfg $OUT/${cfg}_kernel_stats.csv
  echo "stats $cfg done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/raw_fetch_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-boundary --no-shard-rehearsal > /dev/null 2> $OUT/fetch_$cfg.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/raw_write_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-boundary --no-shard-rehearsal > /dev/null 2> $OUT/write_$cfg.err
  python3 tools/pmc_summarize.py pmc $OUT/raw_fetch_$cfg $OUT/raw_write_$cfg $OUT/pmc_$cfg.csv
  echo "pmc $cfg done"
  rm -rf $OUT/raw_stats_$cfg $OUT/raw_fetch_$cfg $OUT/raw_write_$cfg
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $OUT/raw_mfma -- python3 bench.py --config pass --steps 5 --warmup 1 --no-cpu-baseline --no-boundary > /dev/null 2> $OUT/mfma.err
python3 tools/pmc_summarize.py mfma $OUT/raw_mfma $OUT/pmc_mfma_pass.csv
rm -rf $OUT/raw_mfma
echo "mfma done"
