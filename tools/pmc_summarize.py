#!/usr/bin/env python3
"""Summarises rocprofv3 output directories into the small csv files kept under profiles/.

    pmc_summarize.py pmc   <fetch_dir> <write_dir> <out.csv>     per-kernel HBM-side bytes per launch
    pmc_summarize.py stats <stats_dir> <out.csv>                  the *_kernel_stats.csv of a --stats run, copied
    pmc_summarize.py mfma  <dir> <out.csv>                        matrix-core counters of one --pmc pass, per-dispatch averages

FETCH_SIZE / WRITE_SIZE come from separate runs of the same command (they do not fit one pass: MI355X_MICROARCH.md,
"rocprofv3 PMC slots").  rocprofv3 reports both in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so the
corrected traffic is 2 * FETCH + WRITE (same guide, HBM section).  The first line records the hash of the kernel
sources the counters were taken at: bench.py marks the traffic stale when it no longer matches.
"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].strip().replace(",", ";")
                a = acc[name]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    mode = sys.argv[1]
    if mode == "pmc":
        fetch_dir, write_dir, out = sys.argv[2:5]
        fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
        import bench
        with open(out, "w") as fh:
            fh.write(f"# kernel_source_hash={bench.kernel_source_hash()} rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs, KB per dispatch (averages)\n")
            fh.write("kernel,dispatches,FETCH_SIZE_KB_avg_raw,WRITE_SIZE_KB_avg,traffic_bytes_per_launch_corrected\n")
            for k in sorted(fe, key=lambda k: -fe[k][0]):
                f_avg = fe[k][0] / max(fe[k][1], 1)
                w_avg = wr[k][0] / max(wr[k][1], 1) if k in wr else 0.0
                fh.write(f"{k},{fe[k][1]},{f_avg:.1f},{w_avg:.1f},{int(1024 * (2 * f_avg + w_avg))}\n")
    elif mode == "stats":
        d, out = sys.argv[2:4]
        files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if not files:
            raise SystemExit(f"no kernel_stats.csv under {d}")
        shutil.copy(files[0], out)
    elif mode == "mfma":
        d, out = sys.argv[2:4]
        names = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64")
        cols = [per_kernel(d, c) for c in names]
        import bench
        with open(out, "w") as fh:
            fh.write(f"# kernel_source_hash={bench.kernel_source_hash()} rocprofv3 --pmc {' '.join(names)} (one pass, --kernel-trace only), per-dispatch averages\n")
            fh.write("kernel,dispatches," + ",".join(names) + ",mfma_busy_over_sq_busy\n")
            for k in sorted(cols[1], key=lambda k: -cols[1][k][0]):
                avg = [c[k][0] / max(c[k][1], 1) if k in c else 0.0 for c in cols]
                fh.write(f"{k},{cols[1][k][1]}," + ",".join(f"{v:.0f}" for v in avg) + f",{avg[0] / max(avg[1], 1.0):.4f}\n")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
