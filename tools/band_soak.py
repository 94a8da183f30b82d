#!/usr/bin/env python3
"""Soak of the banded reduced solve: random windows (24 .. 110 key frames, short tracks), the two-sided and the one-workgroup
banded factorisations against the general blocked one on the same inputs: same LM schedule, cameras within 1e-8."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(20251005)
ctx = rs.Context(0)
bad = 0
for i in range(n_win):
    n_kf = int(rng.integers(24, 111))
    n_pts = int(rng.integers(400, 3000))
    w = synth.make_ba_window(n_kf=n_kf, n_points=n_pts, run_min=2, run_max=int(rng.integers(4, 11)), config_id=1000 + i)
    dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
    res = {}
    for mode in (0, 2, 1):
        ctx.set_int("ba_band_mode", mode)
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        ctx.prof_begin()
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        prof = ctx.prof_end()
        res[mode] = (s, [t["outcome"] for t in ctx.ba_trace()], dc.cpu().numpy().copy(), "K7b_band_factor" in prof)
    ctx.set_int("ba_band_mode", 0)
    ok = all(res[m][1] == res[1][1] and res[m][0]["iterations"] == res[1][0]["iterations"] for m in (0, 2))
    err = max(np.abs(res[m][2] - res[1][2]).max() for m in (0, 2))
    ok = ok and err < 1e-8 and res[1][0]["usable"] == 1
    bad += 0 if ok else 1
    print(f"window {i}: {n_kf} key frames (n = {6 * int(w['cam_free'].sum())}), {n_pts} points, banded {res[0][3]}: "
          f"{'ok' if ok else 'MISMATCH'} max |dc| {err:.2e}", flush=True)
ctx.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
