#!/usr/bin/env python3
"""Throughput mode: B copies of the cfg-3 window per rs_bundle_adjust_batch call, solves/s by landmarks per item."""
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
w = synth.make_ba_window()
c0, p0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
for B in (8, 32, 128):
    clones = [(c0.clone(), p0.clone()) for _ in range(B)]
    probs = [(bc, w["cam_free"], bp, *dev, w["K"]) for bc, bp in clones]
    for item in (64, 40, 32):
        ctx.set_int("ba_batch_item_landmarks", item)

        def step():
            for bc, bp in clones:
                bc.copy_(c0); bp.copy_(p0)
            torch.cuda.synchronize()
            return ctx.bundle_adjust_batch(probs)

        for _ in range(2):
            out = step()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            out = step()
        dt = (time.perf_counter() - t0) / n
        ctx.prof_begin()
        step()
        p = ctx.prof_end()
        print(f"B={B} item={item}: {B / dt:.0f} solves/s ({1e3 * dt:.2f} ms/call) cost {out[0]['final_cost']:.9g} it {out[0]['iterations']} |",
              " ".join(f"{k[:10]}={1e3 * v[1] / max(v[0], 1):.0f}" for k, v in sorted(p.items())), flush=True)
ctx.close()
