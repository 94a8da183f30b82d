#!/usr/bin/env python3
"""The fused K7 + K8 launch with 3, 2 and 1 resident K8 sets (windows of 9 k .. 30 k landmarks) against the two-launch form:
same LM schedule, same kernels otherwise; cameras within 1e-9, final cost within 1e-10 relative.  (Points are reported, not asserted:
a landmark seen from two nearly parallel rays has a V of condition 1e10 and moves by 1e-4 for a camera difference of 1e-14.)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

ctx = rs.Context(0)
bad = 0
for i, P in enumerate((9000, 10600, 10800, 12000, 15900, 16200, 20000, 30000)):
    w = synth.make_ba_window(n_points=P) if len(sys.argv) > 1 else synth.make_ba_window(n_points=P, config_id=2000 + i)    # (any argument: the benchmark window's seed, whose first steps are rejected)
    dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
    res = {}
    for mode in (2, 1):
        ctx.set_int("ba_fuse_mode", mode)
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        ctx.prof_begin()
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        prof = ctx.prof_end()
        res[mode] = (s, "".join({1: "A", 0: "R", -1: "I", 2: "T"}[t["outcome"]] for t in ctx.ba_trace()), dc.cpu().numpy().copy(), dp.cpu().numpy().copy(),
                     "K78_ba_solve_backsub" in prof, ctx.ba_stats()["handoff_retries"])
    ctx.set_int("ba_fuse_mode", 0)
    ec, ep = np.abs(res[2][2] - res[1][2]).max(), np.abs(res[2][3] - res[1][3]).max()
    dcost = abs(res[2][0]["final_cost"] - res[1][0]["final_cost"]) / res[1][0]["final_cost"]
    nfar = int((np.abs(res[2][3] - res[1][3]).max(axis=1) > 1e-7).sum())
    ok = res[2][1] == res[1][1] and ec < 1e-9 and dcost < 1e-10 and res[2][4] and not res[1][4]
    bad += 0 if ok else 1
    print(f"{P} landmarks: fused {res[2][4]} {res[2][1]} retries {res[2][5]}: {'ok' if ok else 'MISMATCH'} |dc| {ec:.1e} cost {dcost:.1e} |dp| {ep:.1e} ({nfar} of {P} points beyond 1e-7)", flush=True)
ctx.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
