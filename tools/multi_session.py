"""Several independent BA sessions on ONE GPU: one rs context + stream per session, one host thread each (ctypes
releases the GIL during the C call).  Kernels of different sessions overlap on the device (every BA kernel is a latency
chain that leaves most CUs idle), so the aggregate throughput grows without any batching API:
    1 session 2104 BA/s, 2 -> 3307, 4 -> 3331, 8 -> 4703 (MI355X, cfg-3 window, end of round 2; FUSE_MODE=2 keeps K7 + K8 in
    one launch under contention too: 8 -> 3951).
Run from the repo root on a GPU box:  python tools/multi_session.py"""
import sys, importlib, time, threading, numpy as np
sys.path[:0] = ['.']
import torch
pkg = importlib.import_module("racing-slam_amd"); rs, synth = pkg.rsgpu, pkg.synth
w = synth.make_ba_window()
def worker(ctx, stream, n, out, i):
    with torch.cuda.stream(stream):
        ctx.use_stream(stream)
        dc0, dp0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
        args = (ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]))
        dc, dp = dc0.clone(), dp0.clone()
        for _ in range(3): dc.copy_(dc0); dp.copy_(dp0); ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
        stream.synchronize()
        out[i] = ('ready',)
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(n): dc.copy_(dc0); dp.copy_(dp0); s = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
        stream.synchronize()
        out[i] = (time.perf_counter() - t0, s["final_cost"])
import os
FUSE = int(os.environ.get("FUSE_MODE", "0"))          # 0: library default (one launch only when alone), 1: two launches, 2: one launch
for nthreads in (1, 2, 4, 8):
    barrier = threading.Barrier(nthreads)
    ctxs = [rs.Context(0) for _ in range(nthreads)]
    for c in ctxs: c.set_int("ba_fuse_mode", FUSE)
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    out = [None] * nthreads
    n = 40
    ths = [threading.Thread(target=worker, args=(ctxs[i], streams[i], n, out, i)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    tmax = max(o[0] for o in out)
    print(f"fuse_mode {FUSE}: {nthreads} concurrent contexts: {nthreads * n / tmax:8.1f} BA/s aggregate ({1e3 * tmax / n:.3f} ms per BA per context), cost {out[0][1]:.6f}")
    for c in ctxs: c.close()
