#!/usr/bin/env python3
"""Phase stamps of the grouped K2 (build with RS_STAMPS=1): phase ends of one wave of the middle workgroup, relative to
its start, in microseconds: [staged-loads issued, tree staged, gates, traversal, compare, outputs]."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

ctx = rs.Context(0)
w = synth.make_ba_window()
frame, mp = synth.make_match_scene(w, n_keypoints=2000, kdtree_build=rs.kdtree_build)
fv, keep_f = ctx.make_frame_view(frame, pack=True)
mv, keep_m = ctx.make_map_view(mp)
out = ctx.reproj_match(fv, mv)
ctx.synchronize()
buf = (C.c_ulonglong * 8)()
acc = [0.0] * 7
n = 50
for it in range(n + 1):
    ctx.lib.rs_k2_stamps(buf, 1)
    ctx.reproj_match(fv, mv, out=out)
    ctx.synchronize()
    ctx.lib.rs_k2_stamps(buf, 0)
    if it:
        for i in range(1, 7):
            acc[i] += (buf[i] - buf[0]) / 100.0
print("K2 phase ends (us after the wave started):", [round(a / n, 2) for a in acc[1:]])
ctx.close()
