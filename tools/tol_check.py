#!/usr/bin/env python3
"""the non-banded window of test_bundle_adjust_banded_reduced_solve (40 key frames, 4000 landmarks seen over up to
24 frames) is accepted at rtol 1e-5.  Measures what that window needs: GPU run-to-run and GPU-vs-oracle differences of
cameras and points, and the conditioning that explains them — the Jacobi-scaled reduced camera system S of the FINAL state
(numpy; complex-step Jacobians from tests/dense_lm.py), undamped and at the final radius."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth
import pyoracle as O  # noqa: E402
import dense_lm as D  # noqa: E402


def main():
    kw = dict(n_kf=40, n_points=4000, run_min=3, run_max=24, config_id=143)
    w = synth.make_ba_window(**kw)
    rc, rp, osum = O.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    ctx = rs.Context(0)
    dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
    runs = []
    for rep in range(4):
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        runs.append((dc.cpu().numpy(), dp.cpu().numpy(), s))

    def diff(a, b):
        d = np.abs(a - b)
        return float(d.max()), float((d / np.maximum(np.abs(b), 1e-300)).max()), float((d / (1e-8 + np.abs(b))).max())
    print("radius", runs[0][2]["final_radius"], "cost gpu / oracle", runs[0][2]["final_cost"], osum["final_cost"])
    for i in range(1, 4):
        print("run 0 vs run", i, "cams (max abs, max rel, max d / (1e-8 + |x|))", diff(runs[i][0], runs[0][0]), "points", diff(runs[i][1], runs[0][1]))
    print("gpu vs oracle cams", diff(runs[0][0], rc), "points", diff(runs[0][1], rp))
    for name, (cond, lo, hi) in D.reduced_system_condition(w, rc, rp, osum["final_radius"]).items():
        print(f"reduced camera system, Jacobi-scaled, {name}: condition {cond:.3e} (eigenvalues {lo:.3e} .. {hi:.3e})")
    ctx.close()


if __name__ == "__main__":
    main()
