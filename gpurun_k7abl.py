import sys, importlib, os, subprocess
if len(sys.argv) > 1:
    sys.path[:0]=['.', 'oracle']
    pkg = importlib.import_module("racing-slam_amd"); rs, synth = pkg.rsgpu, pkg.synth
    ctx = rs.Context(0)
    out = []
    for n_kf in (8, 20):
        w = synth.make_ba_window(n_kf=n_kf, n_points=2000, run_min=2, run_max=min(10, n_kf), config_id=3)
        dc0, dp0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
        args = (ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]))
        for rep in range(3):
            dc, dp = dc0.clone(), dp0.clone()
            ctx.prof_begin(); ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"]); prof = ctx.prof_end()
        k7 = prof["K7_ba_reduced_solve"]; out.append(1e3*k7[1]/k7[0])
    print(f"dbg={os.environ.get('RS_K7_DEBUG','0'):>3s}: K7(6 steps) {out[0]:6.2f} us, K7(18 steps) {out[1]:6.2f} us, per step {(out[1]-out[0])/12:5.2f} us")
else:
    for m in (0, 64, 3):
        env = dict(os.environ, RS_K7_DEBUG=str(m))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
