#!/usr/bin/env python3
"""cfg-3 bundle adjustment with K7 + K8 as one launch (ba_fuse_mode 2; the default 0 does so when no other solve is in
flight) and as two (1): ms per solve and the
per-scope averages."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

ctx = rs.Context(0)
w = synth.make_ba_window()
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
c0, p0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
for rep in range(2):
    for mode in (1, 2):
        ctx.set_int("ba_fuse_mode", mode)
        for _ in range(10):
            dc.copy_(c0); dp.copy_(p0)
            ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        ctx.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            dc.copy_(c0); dp.copy_(p0)
            s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        ctx.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        ctx.prof_begin()
        for _ in range(20):
            dc.copy_(c0); dp.copy_(p0)
            ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        ctx.synchronize()
        p = ctx.prof_end()
        print("ba_fuse_mode", mode, "ms/solve %.4f" % ms, "final_cost %.9g" % s["final_cost"],
              {k: round(1e3 * v[1] / v[0], 2) for k, v in p.items() if k.startswith("K")})
ctx.close()
