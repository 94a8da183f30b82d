#!/usr/bin/env python3
"""Timeline of the one-launch round (ba_round; build with RS_STAMPS=1), last round of a cfg-3 solve, microseconds on the
100 MHz wall clock relative to K7's start: K7 set 0 — items all counted, accumulators taken, delta_c published; one item
workgroup — started, linearisation + Schur done (counted), saw TAKEN, saw delta_c, back-substitution done."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

ctx = rs.Context(0)
ctx.set_int("ba_fuse_mode", 3)
w = synth.make_ba_window()
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
for rep in range(4):
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
    ctx.synchronize()
    c = ctx.prof_counters(48)
    t0 = c[45]
    rel = lambda v: round((v - t0) / 100.0, 2)      # noqa: E731
    print("K7: start 0, items counted", rel(c[44]), "prologue loaded", rel(c[42]), "taken", rel(c[43]), "delta_c", rel(c[40]),
          "| item: start", rel(c[32]), "counted", rel(c[33]), "saw taken", rel(c[34]), "saw delta_c", rel(c[35]), "done", rel(c[36]),
          "| latest item: arithmetic done", rel(c[38]), "atomics drained", rel(c[37]),
          "| slowest item", (c[26] >> 8) & 0xFFFFFF, "union", c[26] & 0xFF, "us", (c[26] >> 32) / 100.0,
          "fastest", (c[27] >> 8) & 0xFFFFFF, "union", c[27] & 0xFF, "us", (c[27] >> 32) / 100.0,
          "start skew us", (c[28] - c[29]) / 100.0)
ctx.close()
