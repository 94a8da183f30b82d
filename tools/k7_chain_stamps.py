#!/usr/bin/env python3
"""Chain-wave phases of K7 (RS_STAMPS=1 build, workgroup of set 0): ticks of the 100 MHz wall clock (10 ns) per block step, averaged over the launches of one cfg-3 solve:
raw column + fix-up | diagonal block through the scratch | 6x6 L D L^T | panel rows | stores | barrier wait."""
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
w = synth.make_ba_window()
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
for rep in range(2):
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
    ctx.synchronize()
    c = ctx.prof_counters(48)
    st = ctx.ba_stats()
    st["rounds"] = max(int(c[31]), 1)        # launches the stamping workgroup (set 0) went through the loop
    steps = 18 * st["rounds"]
    names = ("fixup", "diag_rt", "factor", "solve", "stores", "barrier")
    print(sys.argv[1:], {nm: round(c[16 + i] / steps, 1) for i, nm in enumerate(names)}, "sum", round(sum(c[16:22]) / steps, 1), "| us per launch: set-up wait", round(c[22] / st["rounds"] / 100, 2), "set-up + loop", round(c[30] / st["rounds"] / 100, 2), "loop end -> backsub done", round(c[25] / st["rounds"] / 100, 2),
          "| K7 body stamps", [round(v / st["rounds"]) for v in c[:8]])
ctx.close()
