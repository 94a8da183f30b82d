#!/usr/bin/env python3
"""A/B of context knobs on the cfg-3 bundle adjustment: us per solve + per-scope kernel averages per setting.

    knob_time.py name=v1,v2,... [name2=...] [--window cfg3|cfg5|clean] [--reps N]

e.g.  knob_time.py ba_s_replicas=1,2,4,8        (every combination of the listed values is run, twice, interleaved)"""
import importlib
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

knobs, window, reps = [], "cfg3", 200
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--window":
        window = args.pop(0)
    elif a == "--reps":
        reps = int(args.pop(0))
    else:
        k, v = a.split("=")
        knobs.append((k, [int(x) for x in v.split(",")]))
ctx = rs.Context(0)
kw = {"cfg3": dict(), "cfg5": dict(n_kf=100, n_points=80000, config_id=5)}.get(window, dict(outlier_frac=0.0, pixel_noise=0.3, rot_noise_deg=0.2, config_id=23))
w = synth.make_ba_window(**kw)
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
c0, p0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])


def solve():
    dc.copy_(c0); dp.copy_(p0)
    return ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])


for rep in range(2):
    for combo in itertools.product(*[v for _, v in knobs]):
        for (k, _), v in zip(knobs, combo):
            ctx.set_int(k, v)
        for _ in range(10):
            solve()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            s = solve()
        ctx.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / reps
        ctx.prof_begin()
        for _ in range(20):
            solve()
        ctx.synchronize()
        p = ctx.prof_end()
        tag = " ".join(f"{k}={v}" for (k, _), v in zip(knobs, combo))
        print(f"{tag}: {us:.1f} us/solve cost {s['final_cost']:.9g} it {s['iterations']} |",
              " ".join(f"{k[:12]}={1e3 * v[1] / max(v[0], 1):.1f}" for k, v in sorted(p.items())), flush=True)
ctx.close()
