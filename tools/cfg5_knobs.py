#!/usr/bin/env python3
"""cfg 5 (100 key frames / 80 k landmarks) solve time under context knobs given as name=value arguments."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth
import torch
ctx = rs.Context(0)
for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a):
    ctx.set_int(k, int(v))
w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5)
c0, p0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
dc, dp = c0.clone(), p0.clone()
for _ in range(2):
    dc.copy_(c0); dp.copy_(p0)
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(5):
        dc.copy_(c0); dp.copy_(p0)
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 5)
ctx.prof_begin()
dc.copy_(c0); dp.copy_(p0)
s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
prof = ctx.prof_end()
print(sys.argv[1:], f"{1e3 * best:.3f} ms per solve, iterations {s['iterations']} cost {s['final_cost']:.4f}", {k: (v[0], round(1e3 * v[1] / max(v[0], 1), 1)) for k, v in prof.items()})
ctx.close()
