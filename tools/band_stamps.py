#!/usr/bin/env python3
"""Phase times of the banded factorisation (ba_band_factor; build with RS_STAMPS=1) on cfg 5: microseconds per launch."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth
ctx = rs.Context(0)
w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5)
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
for rep in range(2):
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
    ctx.synchronize()
    c = ctx.prof_counters(48)
    names = ["load/prefetch", "A diag 16x16", "B forward subst", "C rank-16 update", "store factor", "trailing MFMA", "shift window"]
    for side, base in ((0, 16), (1, 40)):       # the two sides of the two-sided form (side 1 is idle in "ba_band_mode" 2)
        print("side", side, {n: round(v / 100.0 / max(s["iterations"], 1), 2) for n, v in zip(names, c[base:base + 7])},
              "total", round(sum(c[base:base + 7]) / 100.0 / s["iterations"], 1))
ctx.close()
