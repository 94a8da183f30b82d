#!/usr/bin/env python3
"""Phases of ba_band_factor (RS_STAMPS=1 build): us per launch and side on the cfg-5 window, from wall-clock stamps of thread 0:
prologue / panels (8-column steps) / trailing updates; inside the panels, thread 0 (a chain wave): wait for the diagonal sub-block,
factor + row solve, wait for the step's multipliers, fix-up; thread 64 (a tile wave): its two waits, update + stores + fetch."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

ctx = rs.Context(0)
for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a):
    ctx.set_int(k, int(v))
w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5)
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
names = ("prologue", "panel", "trailing")
inner = ("c:wait_subblock", "c:factor+solve", "c:wait_step", "c:fixup", "t:waits", "t:after_step", "t:loads+mfma", "t:live+mask", "-", "t:publish")
for rep in range(2):
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    r = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
    ctx.synchronize()
    c = ctx.prof_counters(64)
    rounds = max(int(ctx.ba_stats().get("rounds", 10)), 1)
    for side, base, ib in ((0, 16, 0),):
        print("side", side, {nm: round(c[base + i] / rounds / 100, 2) for i, nm in enumerate(names)}, {nm: round(c[ib + i] / rounds / 100, 2) for i, nm in enumerate(inner)},
              "sum", round(sum(c[base:base + 3]) / rounds / 100, 1), "us per launch;", rounds, "rounds")
ctx.close()
