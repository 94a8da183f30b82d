import sys, importlib, time, numpy as np
sys.path[:0]=['.', 'oracle']
pkg = importlib.import_module("racing-slam_amd"); rs, synth = pkg.rsgpu, pkg.synth
ctx = rs.Context(0)
w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5)
dc0, dp0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
args = (ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]))
for rep in range(2):
    dc, dp = dc0.clone(), dp0.clone()
    ctx.prof_begin(); t0=time.time(); s = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"]); dt=time.time()-t0; prof = ctx.prof_end()
    print('BA', dt*1e3, 'ms', s)
    for k,v in sorted(prof.items()): print('  ', k, v[0], round(1e3*v[1]/max(v[0],1),1), 'us')
