#!/usr/bin/env python3
"""In-kernel phase timers of K5 / K7 (build with RS_STAMPS=1): cycles accumulated by one workgroup per launch."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth

ctx = rs.Context(0)
ctx.set_int("ba_fuse_mode", 1)      # K7's stamps belong to the two-launch form (the fused launch: tools/k78_stamps.py)
big = "cfg5" in sys.argv[1:]
w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5) if big else synth.make_ba_window()
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
for ns in ((1,) if big else (1, 3)):
    ctx.set_int("ba_speculative_sets", ns)
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
    ctx.synchronize()
    c = ctx.prof_counters(16)
    st = ctx.ba_stats()
    print("ns", ns, "rounds", st, "K7 phases (cycles):", c[:8])
    names = ("union+loads", "linearise(pass 1)", "block inverse", "tile zero/Y", "Y -> tile + barrier", "SYRK + scatter", "epilogue", "-")
    print("   K5 phases of the middle workgroup, cycles per launch:", {n: int(v / max(st["rounds"], 1)) for n, v in zip(names, c[8:16])})
ctx.close()
