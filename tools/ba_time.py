#!/usr/bin/env python3
"""Times rs_bundle_adjust alone on two 20-KF windows: the benchmark window (cfg 3: 6 of its 10 LM steps are rejected)
and a clean one on which every step is accepted, for 1 / 2 / 3 speculative radii per round."""
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
for name, kw in (("cfg3", dict()), ("clean", dict(outlier_frac=0.0, pixel_noise=0.3, rot_noise_deg=0.2, config_id=23))):
    w = synth.make_ba_window(**kw)
    c0, p0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
    dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
    for ns in (1, 2, 3):
        ctx.set_int("ba_speculative_sets", ns)
        dc, dp = c0.clone(), p0.clone()
        for _ in range(3):
            dc.copy_(c0); dp.copy_(p0)
            s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            dc.copy_(c0); dp.copy_(p0)
            s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        pat = "".join({1: "A", 0: "R", -1: "I", 2: "T"}[t["outcome"]] for t in ctx.ba_trace())
        print(f"{name} ns={ns}: {1e6 * dt:.1f} us per solve, {pat}, final cost {s['final_cost']:.3f}")
ctx.close()
