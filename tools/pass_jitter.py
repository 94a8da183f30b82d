#!/usr/bin/env python3
"""Wall time of every single benchmark pass (synchronised), to find stalls that an average hides."""
import importlib, os, sys, time, argparse, io, contextlib, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
args = argparse.Namespace(gpus=1, steps=1, warmup=0, config="pass", no_cpu_baseline=True, boundary=False, cpu_seconds=1.0)
e = bench.setup(args)
ts = []
orig_timed = bench.timed
def timed(e_, fn, steps, warmup):
    for i in range(400):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sum(ts)
bench.timed = timed
bench.profiled = lambda e_, fn, steps: {}
with contextlib.redirect_stdout(io.StringIO()):
    try:
        bench.bench_pass(e, args)
    except Exception as ex:
        print("bench_pass aborted after timing:", ex, file=sys.stderr)
import statistics
print("median %.1f us" % (1e6 * statistics.median(ts)))
big = [(i, round(1e3 * t, 2)) for i, t in enumerate(ts) if t > 3 * statistics.median(ts)]
print("passes slower than 3x median (index, ms):", big)
print("gc counts", gc.get_count(), "gc stats", [s["collections"] for s in gc.get_stats()])
