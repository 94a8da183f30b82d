#!/usr/bin/env python3
"""Times rs_refine_pose (tracking, once per frame) on 2000 observations."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth
import numpy as np, torch
ctx = rs.Context(0)
w = synth.make_ba_window(n_kf=4, n_points=2000, run_min=4, run_max=4, config_id=21, outlier_frac=0.05)
sel = np.flatnonzero(w["obs_cam"] == 3)
obs_pt = np.repeat(np.arange(len(w["points"])), np.diff(w["obs_ptr"]))[sel]
pts = ctx.dev(w["points_true"][obs_pt]); uv = ctx.dev(w["obs_uv"][sel])
cam = w["cams"][3]
for _ in range(5):
    c, s = ctx.refine_pose(cam, pts, uv, w["K"])
ctx.synchronize()
t0 = time.perf_counter(); n = 200
for _ in range(n):
    c, s = ctx.refine_pose(cam, pts, uv, w["K"])
dt = (time.perf_counter() - t0) / n
ctx.prof_begin()
for _ in range(50):
    ctx.refine_pose(cam, pts, uv, w["K"])
pk = ctx.prof_end()
print({k: round(1e3 * v[1] / v[0], 1) for k, v in pk.items()})
print("refine_pose: %d observations, %d iterations, %.1f us per call (host round trip included)" % (len(sel), s["iterations"], 1e6 * dt))
ctx.close()
