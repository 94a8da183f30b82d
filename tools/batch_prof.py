#!/usr/bin/env python3
"""A few rs_bundle_adjust_batch calls of B cfg-3 windows (for rocprofv3 --kernel-trace --stats): batch_prof.py [B]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth
import torch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = rs.Context(0)
w = synth.make_ba_window()
c0, p0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
clones = [(c0.clone(), p0.clone()) for _ in range(B)]
probs = [(bc, w["cam_free"], bp, *dev, w["K"]) for bc, bp in clones]
for _ in range(4):
    for bc, bp in clones:
        bc.copy_(c0); bp.copy_(p0)
    torch.cuda.synchronize()
    out = ctx.bundle_adjust_batch(probs)
print(out[0]["final_cost"], out[0]["iterations"])
ctx.close()
