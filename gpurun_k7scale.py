import sys, importlib
sys.path[:0]=['.', 'oracle']
import numpy as np
pkg = importlib.import_module("racing-slam_amd"); rs, synth = pkg.rsgpu, pkg.synth
ctx = rs.Context(0)
for n_kf in (4, 6, 8, 11, 14, 17, 20, 23):
    w = synth.make_ba_window(n_kf=n_kf, n_points=2000, run_min=2, run_max=min(10, n_kf), config_id=3)
    dc0, dp0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
    args = (ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]))
    for rep in range(3):
        dc, dp = dc0.clone(), dp0.clone()
        ctx.prof_begin()
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
        prof = ctx.prof_end()
    k7 = prof["K7_ba_reduced_solve"]; k5 = prof["K5_ba_schur_mfma"]; k8 = prof["K8_ba_backsub_cost"]
    print(f"n_kf={n_kf:3d} n={6*(n_kf-2):4d} steps={n_kf-2:3d}  K7 {1e3*k7[1]/k7[0]:7.2f} us  K5 {1e3*k5[1]/k5[0]:7.2f} us  K8 {1e3*k8[1]/k8[0]:6.2f} us  iters {s['iterations']}")
