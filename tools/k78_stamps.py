#!/usr/bin/env python3
"""Timeline of the fused K7 + K8 launch (build with RS_STAMPS=1), last round of a cfg-3 solve, microseconds relative to
K7's start: accumulators taken, delta_c published; first K8 workgroup: started, ready to wait, saw the word, operands
staged, arithmetic done."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

ctx = rs.Context(0)
w = synth.make_ba_window()
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
for rep in range(3):
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
    ctx.synchronize()
    c = ctx.prof_counters(48)
    t0 = c[42]
    rel = lambda v: round((v - t0) / 100.0, 2)
    print("K7: start 0, accumulators taken", rel(c[43]), "delta_c published", rel(c[40]), "| K8 wg0: start", rel(c[32]), "ready", rel(c[33]), "saw", rel(c[34]),
          "staged", rel(c[35]), "computed", rel(c[36]))
ctx.close()
