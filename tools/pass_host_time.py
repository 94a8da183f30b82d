#!/usr/bin/env python3
"""Host time of every call of the benchmark pass (perf_counter around the calls, no extra synchronisation): where the
interpreter, not the GPU, sets the pace.  Prints the median of 200 passes per call, microseconds."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

e = bench.setup(argparse.Namespace(gpus=1))
one_pass, _, meta = bench.build_pass(e, 1)
ctx, rs = e.ctx, e.rs
names, calls = [], {}
orig = {}


def wrap(obj, name, tag=None):
    fn = getattr(obj, name)
    tag = tag or name
    orig[(obj, name)] = fn

    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        calls.setdefault(tag, []).append(time.perf_counter() - t0)
        return r
    setattr(obj, name, w)


for n in ("match_descriptors", "reproj_match", "triangulate_matches", "triangulate_tracks", "bundle_adjust", "ba_cameras",
          "reanchor_points_host_poses"):
    wrap(ctx, n)
for n in ("build_local_window", "unpack_poses"):
    wrap(rs, n)
one_pass.prepare(260)
for _ in range(20):
    one_pass()
calls.clear()
t0 = time.perf_counter()
for _ in range(200):
    one_pass()
e.torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 200
for k, v in calls.items():
    v.sort()
    per_pass = len(v) / 200
    print(f"{k:28s} {1e6 * v[len(v) // 2]:8.1f} us x {per_pass:.0f}")
print(f"pass {1e6 * total:.1f} us; sum of medians {1e6 * sum(sorted(v)[len(v) // 2] * len(v) / 200 for v in calls.values()):.1f} us")
