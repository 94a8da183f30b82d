#!/usr/bin/env python3
"""Host-side cost of every API call of the benchmark pass (time to ENQUEUE, no synchronisation): the front-end kernels
last 8-35 us each, so a pass is only as fast as the host can feed them (python bench.py on a loaded host drops from
~1450 to ~800 passes/s with unchanged kernel times)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import argparse, numpy as np, torch
args = argparse.Namespace(gpus=1, steps=50, warmup=5, config="pass", no_cpu_baseline=True, boundary=False, cpu_seconds=1.0)
e = bench.setup(args)
ctx, rs = e.ctx, e.rs
calls = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); calls.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
for n in ("match_descriptors", "reproj_match", "triangulate_matches", "bundle_adjust", "ba_cameras", "reanchor_points"):
    wrap(ctx, n)
for n in ("build_local_window", "unpack_poses"):
    wrap(rs, n)
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.bench_pass(e, args)
import json
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
for k, v in calls.items():
    v = sorted(v)
    print("%-22s n=%d median %.1f us p90 %.1f" % (k, len(v), 1e6 * v[len(v) // 2], 1e6 * v[int(len(v) * 0.9)]))
