import importlib, sys, ctypes as C
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("racing-slam_amd")
rs, synth = pkg.rsgpu, pkg.synth
ctx = rs.Context(0)
w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5)
dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
o = rs.default_options(); o.max_num_iterations = 2
s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"], options=o)
ctx.synchronize()
buf = (C.c_ulonglong * 64)()
print(rs.load().rs_debug_read(buf))
t = list(buf)
print([ (i, (t[i]-t[0])*10) for i in range(17) if t[i]])
