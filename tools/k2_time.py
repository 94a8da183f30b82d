#!/usr/bin/env python3
"""Times K2 (rs_reproj_match) alone on the benchmark scene (20 KF / 10 k landmarks / 2000 keypoints)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth
import torch  # noqa: E402

ctx = rs.Context(0)
w = synth.make_ba_window()
frame, mp = synth.make_match_scene(w, n_keypoints=2000, kdtree_build=rs.kdtree_build)
fv, keep_f = ctx.make_frame_view(frame, pack=True)
mv, keep_m = ctx.make_map_view(mp)
for mode in (0, 1):                                  # eight lanes per map point / one lane per point
    ctx.set_int("k2_mode", mode)
    out = ctx.reproj_match(fv, mv)
    for _ in range(20):
        ctx.reproj_match(fv, mv, out=out)
    ctx.synchronize()
    ctx.prof_begin()
    for _ in range(200):
        ctx.reproj_match(fv, mv, out=out)
    ctx.synchronize()
    p = ctx.prof_end()
    print("k2_mode", mode, {k: round(1e3 * v[1] / v[0], 2) for k, v in p.items()}, "matches", int(out["count"].item()))
ctx.close()
