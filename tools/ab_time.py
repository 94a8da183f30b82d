#!/usr/bin/env python3
"""A/B timing of one cfg-3 solve (cfg5: the 100-KF / 80k-landmark window) for the library named by RS_LIB (build.py, RS_VARIANT):
run once per variant in the same gpurun call.  Prints the per-kernel HIP-event times of the solve."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth
import torch  # noqa: E402

ctx = rs.Context(0)
for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("points=")):
    ctx.set_int(k, int(v))
big = "cfg5" in sys.argv[1:]
npts = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("points=")]
w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5) if big else (synth.make_ba_window(n_points=npts[0]) if npts else synth.make_ba_window())
c0, p0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
dc, dp = c0.clone(), p0.clone()
for _ in range(5):
    dc.copy_(c0); dp.copy_(p0)
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    n = 10 if big else 40
    for _ in range(n):
        dc.copy_(c0); dp.copy_(p0)
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / n)
pat = "".join({1: "A", 0: "R", -1: "I", 2: "T"}[t["outcome"]] for t in ctx.ba_trace())
ctx.prof_begin()
dc.copy_(c0); dp.copy_(p0)
s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
prof = ctx.prof_end()
print(os.environ.get("RS_LIB", "librsgpu.so"), sys.argv[1:], f"{1e6 * best:.1f} us per solve {pat} cost {s['final_cost']:.6f}",
      {k: (v[0], round(1e3 * v[1] / max(v[0], 1), 1)) for k, v in prof.items()})
ctx.close()
