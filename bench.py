#!/usr/bin/env python3
"""bench.py — the hot path of Racing-SLAM on MI355X, measured.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config pass|cfg2|cfg3|cfg4|cfg5]

--config pass (default; BASELINE.json's metric "match+triangulate+local-BA passes/sec"):
  one STEP = one pass of the hot path over one batch of synthetic input, everything resident in HBM before the
  timed region, reproducing the reference's call pattern per key frame (SURVEY.md §8 a-callers):
    (1) brute-force 2-NN Hamming + Lowe/threshold match, 2000 x 2000 ORB rows         cfg 2, MapMatcher::match_descriptors
    (2) TWO reprojection-gated matches of the new frame (src/Tracker.cpp:232-248):
        match_key_frame (points observed by the last key frame), then match_map (whole map, with the keypoints
        and points matched by the first call taken out)
    (3) two-view DLT triangulation + gates of the accepted pairs                       cfg 2, triangulate_points
    (4) Mapper::bundle_adjust (src/Mapper.cpp:364-394): build_local_window (host), the 10-iteration local BA
        on 20 KF x 10k landmarks x ~60k observations (cfg 3), read-back + unpack of the refined poses, rigid
        re-anchoring of the single-observation points
  N > 1: every rank runs (1)-(3) on its own frame pair and owns a 10k-landmark shard of a 20-KF window with
  N x 10k landmarks; the BA all-reduces the reduced camera system over RCCL each LM round (weak scaling).
--config cfg2 / cfg3: the two single-GPU configurations alone (match + triangulate of one pair; the local BA alone).
--config cfg4: 64 key-frame pairs x 2000 keypoints, batched match + batched triangulation; N ranks take 64/N pairs
  each, no collective (strong scaling).   --config cfg5: 100-KF / 80k-landmark BA; N ranks take 80k/N landmarks each,
  RCCL all-reduce of the 588 x 588 reduced system per LM round (strong scaling).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the library's stream) and
`cpu_baseline` (the CPU restatement built -O3 -march=native on this host, 1 thread; the all-cores OpenMP figure
beside it; median / p10 / p90 over the repetitions).
"""
import argparse
import gc
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == fp64 MFMA peak (public spec)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md
VALU_INT_PEAK_TOPS = 39.3  # 256 CU x 64 lanes x 2.4 GHz 32-bit integer ops (K1's real ceiling)
METRIC = "match+triangulate+local-BA passes/sec @ 2k kpts/frame, 20-KF x 10k-pt window"

PMC_NAMES = {"K5_ba_schur_mfma": "ba_schur_mfma", "K7_ba_reduced_solve": "ba_reduced_solve_lds", "K78_ba_solve_backsub": "ba_solve_backsub",
             "K8_ba_backsub_cost": "ba_backsub_cost4", "K1_hamming_knn2": "k1_hamming_knn2",
             "K1b_merge_filter": "k1_merge_filter", "K2_reproj_match": "k2_reproj_match",
             "K4_triangulate_dlt": "k4_triangulate", "K7_ba_reduced_solve_blocked": "ba_big_update"}


def kernel_source_hash():
    """Hash of the kernel sources: PMC files record it, so that a stale counter file is recognisable."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "racing-slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(config):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes of THIS benchmark command (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, unit / gfx950 corrections as MI355X_MICROARCH.md prescribes; see
    profiles/README.md).  Counters cannot be collected from inside bench.py: the file is the record of the last
    collection, with the kernel-source hash it was taken at."""
    path = os.path.join(ROOT, "profiles", f"round2_pmc_{config}.csv")
    out, src_hash = {}, None
    try:
        with open(path) as fh:
            for line in fh:
                if line.startswith("#"):
                    if "kernel_source_hash=" in line:
                        src_hash = line.strip().split("kernel_source_hash=")[1].split()[0]
                    continue
                f = line.strip().rsplit(",", 4)          # kernel names may contain commas (template arguments)
                if len(f) == 5 and f[0] != "kernel":
                    try:
                        out[f[0]] = float(f[4])
                    except ValueError:
                        pass
    except OSError:
        return {}, None, None
    return out, os.path.relpath(path, ROOT), (src_hash == kernel_source_hash())


def algorithmic_work(w, n_free):
    """Per-LM-iteration algorithmic work of the BA, unpadded and sparse."""
    k = np.diff(w["obs_ptr"]).astype(np.float64)
    M, P, C = float(k.sum()), float(len(k)), float(len(w["cams"]))
    n = 6.0 * n_free
    return dict(M=M, P=P, C=C, n=n, lin_flops=500.0 * M,
                schur_flops=float(np.sum(50 + 108 * k + 216 * k * (k + 1) / 2 + 72 * k)),
                solve_flops=n ** 3 / 3, backsub_flops=410.0 * M + 50.0 * P, lin_bytes=16 * M + 2 * 24 * P + 2 * 48 * C + 8 * n * n,
                cost_bytes=16 * M + 24 * P + 48 * C)


class Env:
    pass


def setup(args):
    import torch
    import torch.distributed as dist
    e = Env()
    e.torch, e.dist = torch, dist
    e.pkg = importlib.import_module("racing-slam_amd")
    e.rs, e.synth = e.pkg.rsgpu, e.pkg.synth
    e.world = int(os.environ.get("WORLD_SIZE", "1"))
    e.rank = int(os.environ.get("RANK", "0"))
    e.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != e.world and e.world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(e.local_rank)
    if e.world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", e.local_rank))
    e.ctx = e.rs.Context(e.local_rank)
    return e


def attach_comm(e):
    """RCCL communicator for the landmark-sharded BA (the only stage with an exchange step)."""
    if e.world > 1:
        torch, dist = e.torch, e.dist
        uid = torch.zeros(128, dtype=torch.uint8, device=e.ctx.device)
        if e.rank == 0:
            uid.copy_(torch.frombuffer(bytearray(e.rs.Context.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        e.ctx.comm_init(bytes(uid.cpu().numpy().tobytes()), e.world, e.rank)


def timed(e, fn, steps, warmup):
    """W untimed + exactly K timed steps between barrier + synchronize; MAX over ranks."""
    torch, dist = e.torch, e.dist

    def barrier():
        if e.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        fn()
    # The interpreter's cyclic garbage collector is kept out of the timed region (as timeit does): a full collection
    # over the ~10^6 objects a process that imported torch holds takes ~50 ms — 70 passes' worth — and fires once
    # every few thousand small allocations (tools/pass_jitter.py: one 54 ms pass among 400).
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    if e.world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=e.ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def profiled(e, fn, steps):
    """Per-kernel HIP-event times in a separate instrumented repetition (event records perturb the timed region)."""
    e.ctx.prof_begin()
    for _ in range(steps):
        fn()
    prof = e.ctx.prof_end()
    return {k: dict(launches=v[0], avg_us=1e3 * v[1] / max(v[0], 1), total_ms=v[1]) for k, v in prof.items()}


def cpu_measure(fn, budget_s, min_reps=20, warm=1):
    """median / p10 / p90 of fn's wall time: `min_reps` repetitions, fewer when the time budget runs out first
    (never fewer than 3)."""
    for _ in range(warm):
        fn()
    ts = []
    t_end = time.perf_counter() + budget_s
    while len(ts) < min_reps and (len(ts) < 3 or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    a = np.array(ts)
    return dict(median_s=float(np.median(a)), p10_s=float(np.percentile(a, 10)), p90_s=float(np.percentile(a, 90)),
                reps=len(ts))


def C_omp_set_threads(n):
    """omp_set_num_threads on the OpenMP runtime the baseline library linked (libgomp)."""
    import ctypes
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    return True


def cpu_baseline(cpu_fn, units_per_call, unit, what, budget_s):
    """The oracle's sources rebuilt as a BASELINE (-O3 -march=native on this host; never the checker binary):
    1 thread, and all cores with OpenMP (mirrors Ceres' num_threads = hardware_concurrency, src/Optimization.cpp:122-132)."""
    import pyoracle as O
    model, ncpu, _ = O.cpu_model()
    ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else ncpu
    ncore = max(1, min(ncore, 16))      # a one-GPU box's CPU share is 16 cores, whatever the affinity mask says
    out = {}
    try:
        O.use_baseline("fast")
        m = cpu_measure(lambda: cpu_fn(O), budget_s * 0.6)
        out["one"] = dict(value=units_per_call / m["median_s"], unit=unit, cores=1, kind="port",
                          p10=units_per_call / m["p90_s"], p90=units_per_call / m["p10_s"], repetitions=m["reps"],
                          cpu_model=model, nproc=ncpu,
                          sample=what + "; oracle sources built gcc -O3 -march=native on this host (64-bit popcount, "
                                        "Ceres-style jets), 1 thread; faithful restatement, not the reference binary")
        # The box's CPU share is a cgroup quota, not the affinity mask: more OpenMP threads than the quota only
        # contend.  Try the full share, half and a quarter (passive waiting) and report the best.
        os.environ["OMP_WAIT_POLICY"] = "passive"
        lib_omp = O.use_baseline("fast_omp")
        best = None
        for nt in sorted({ncore, max(1, ncore // 2), max(1, ncore // 4)}, reverse=True):
            try:
                lib_omp_set = C_omp_set_threads(nt)
            except OSError:
                lib_omp_set = False
            mm = cpu_measure(lambda: cpu_fn(O), budget_s * 0.4 / 3, min_reps=7 if best else 20)
            if best is None or mm["median_s"] < best[0]["median_s"]:
                best = (mm, nt)
        m, ncore = best
        del lib_omp, lib_omp_set
        out["all"] = dict(value=units_per_call / m["median_s"], unit=unit, cores=ncore, kind="port",
                          p10=units_per_call / m["p90_s"], p90=units_per_call / m["p10_s"], repetitions=m["reps"],
                          cpu_model=model, nproc=ncpu,
                          sample=what + "; the same build + OpenMP over queries / map points / correspondences / "
                                        "observations / landmark blocks")
    finally:
        O.use_baseline(None)
    return out


def roofline_entry(name, bound, work_amount, per_kernel, pmc, note=None):
    if name not in per_kernel:
        return None
    t = per_kernel[name]["avg_us"] * 1e-6
    regime = None
    if bound == "latency":
        # a dependency chain in one workgroup per radius: priced against the fp64 peak as the contract's "mfma" bound asks,
        # but the regime says what actually limits it
        bound, regime = "mfma", "latency chain (one workgroup per speculative radius): neither the MFMA nor the HBM roof applies"
    if bound == "mfma":
        peak, unit, ach = FP64_PEAK_TFLOPS, "TFLOP/s", work_amount / t / 1e12
    elif bound == "valu-int":
        peak, unit, ach = VALU_INT_PEAK_TOPS, "Tops/s", work_amount / t / 1e12
    else:
        peak, unit, ach = HBM_PEAK_GBS, "GB/s", work_amount / t / 1e9
    r = dict(kernel=name, bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak,
             avg_launch_us=per_kernel[name]["avg_us"], traffic=pmc.get(PMC_NAMES.get(name)),
             algorithmic_work_per_launch=work_amount)
    if regime:
        r["regime"] = regime
    if note:
        r["note"] = note
    return r


def finish(e, args, line):
    if e.rank == 0:
        print(json.dumps(line))
    e.ctx.close()
    if e.world > 1:
        e.dist.destroy_process_group()


def boundary_timings(reps=30):
    """The CALLER's cost per interface call (host objects in -> host results out) through the C++ host mirror, i.e.
    what the replaced translation units of INTEGRATION.md pay including flattening the pointer graph, the staging-pool
    upload and the read-back.  `value` never includes this (inputs resident); it is reported beside it."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host_cpp", "bench_boundary.bin")
    if not os.path.exists(exe):
        return {"error": "tests/host_cpp/bench_boundary.bin not built (run __graft_entry__.build())"}
    try:
        r = subprocess.run([exe, str(reps)], capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "bench_boundary.bin failed", "stderr": r.stderr[-400:], "stdout": r.stdout[-400:]}
        out = json.loads(lines[-1])
        out["what"] = ("end-to-end microseconds per call of the host mirror (racing-slam_amd/host/slam_host.cpp) on a 20-KF / "
                       "2000-keypoint scene: marshal + upload (staging pool) + kernels + read-back; median / p10 / p90")
        return out
    except Exception as ex:      # noqa: BLE001
        return {"error": repr(ex)}


# =============================================================================================== pass
def build_pass(e):
    """Inputs of one pass, resident in HBM, and the closures that run it on the GPU and on the CPU."""
    ctx, rs, synth, torch = e.ctx, e.rs, e.synth, e.torch
    world, rank = e.world, e.rank
    pair = synth.make_pair(2, seed_stream=rank)                         # cfg 2, this rank's frame pair
    window_all = synth.make_ba_window(n_kf=20, n_points=10000 * world)  # cfg 3 (x N landmarks)
    window = synth.shard_ba_by_landmark(window_all, world, rank) if world > 1 else window_all
    shard_for_match = dict(window_all)
    if world > 1:
        lo, hi = window["point_range"]
        shard_for_match.update(points=window_all["points"][lo:hi], points_true=window_all["points_true"][lo:hi],
                               obs_ptr=window["obs_ptr"], obs_cam=window["obs_cam"], obs_uv=window["obs_uv"])
    frame, mp = synth.make_match_scene(shard_for_match, n_keypoints=2000, kdtree_build=rs.kdtree_build)
    n_kf = len(window["cams"])
    P = len(mp["positions"])

    # (2) the two calls of Tracker (src/Tracker.cpp:232-248).  match_key_frame only runs the points the last key
    # frame observes (src/MapMatcher.cpp:169); Frame::add_map_match then marks keypoints and points as matched, so
    # match_map sees them taken out (:53, :81).  The second view is fixed at set-up from the first call's result.
    obs_pt = np.repeat(np.arange(P), np.diff(mp["obs_ptr"]))
    seen_by_last = np.zeros(P, bool)
    seen_by_last[obs_pt[mp["obs_kf"] == n_kf - 1]] = True
    mp_a = dict(mp, eligible=(mp["eligible"].astype(bool) & seen_by_last).astype(np.uint8))
    fv_a, keep_fa = ctx.make_frame_view(frame, pack=True)
    mv_a, keep_ma = ctx.make_map_view(mp_a)
    r_a = ctx.reproj_match(fv_a, mv_a)
    cnt_a = int(r_a["count"].cpu()[0])
    kp_a = r_a["match_kp"].cpu().numpy()[:cnt_a]
    pt_a = r_a["match_point"].cpu().numpy()[:cnt_a]
    frame_b = dict(frame, kp_matched=frame["kp_matched"].copy())
    frame_b["kp_matched"][kp_a] = 1
    elig_b = mp["eligible"].copy()
    elig_b[pt_a] = 0
    mp_b = dict(mp, eligible=elig_b)
    fv_b, keep_fb = ctx.make_frame_view(frame_b, pack=True)
    mv_b, keep_mb = ctx.make_map_view(mp_b)
    r_b = ctx.reproj_match(fv_b, mv_b)

    # (4) build_local_window input (host): covisibility CSR of the window (frame -> points, point -> observers)
    full_obs_pt = np.repeat(np.arange(len(window["points"])), np.diff(window["obs_ptr"])).astype(np.int32)
    order = np.argsort(window["obs_cam"], kind="stable")
    frame_pt = full_obs_pt[order]
    frame_ptr = np.zeros(n_kf + 1, np.int32)
    frame_ptr[1:] = np.cumsum(np.bincount(window["obs_cam"], minlength=n_kf))
    lw_args = (n_kf, n_kf - 1, 20, 0, frame_ptr, frame_pt, window["obs_ptr"].astype(np.int32), window["obs_cam"].astype(np.int32))
    # single-observation points (excluded from the BA, re-anchored afterwards): 2000, spread over the free frames
    rng = np.random.default_rng(77 + rank)
    free_idx = np.flatnonzero(window["cam_free"])
    n_single = 2000
    single_frame = rng.choice(free_idx, n_single).astype(np.int32)
    single_pos = (window["points_true"][rng.integers(0, len(window["points_true"]), n_single)]
                  + rng.normal(0, 0.05, (n_single, 3))).astype(np.float32)
    poses_before = np.stack([rs.unpack_pose(c) for c in window["cams"]]).reshape(-1, 16).astype(np.float32)

    nq, nt = len(pair["desc2"]), len(pair["desc1"])
    d = dict(q=ctx.dev(pair["desc2"]), t=ctx.dev(pair["desc1"]), kp1=ctx.dev(pair["kp1"]), kp2=ctx.dev(pair["kp2"]),
             poses=ctx.dev(pair["poses"]), cams0=ctx.dev(window["cams"]), pts0=ctx.dev(window["points"]),
             optr=ctx.dev(window["obs_ptr"]), ocam=ctx.dev(window["obs_cam"]), ouv=ctx.dev(window["obs_uv"]),
             single0=ctx.dev(single_pos), single_frame=ctx.dev(single_frame), before=ctx.dev(poses_before))
    # working copies of what a pass modifies (cameras, points, re-anchored points), reset by ONE device copy per pass:
    # the reset is bookkeeping of the benchmark, not part of the path
    nb_c, nb_p, nb_s = d["cams0"].numel() * 8, d["pts0"].numel() * 8, d["single0"].numel() * 4
    off_p, off_s = (nb_c + 255) // 256 * 256, ((nb_c + 255) // 256 * 256) + (nb_p + 255) // 256 * 256
    state0 = torch.zeros(off_s + nb_s, dtype=torch.uint8, device=d["cams0"].device)
    state = torch.zeros_like(state0)

    def views(buf):
        return (buf[0:nb_c].view(torch.float64).view(d["cams0"].shape), buf[off_p:off_p + nb_p].view(torch.float64).view(d["pts0"].shape),
                buf[off_s:off_s + nb_s].view(torch.float32).view(d["single0"].shape))
    for dst, src in zip(views(state0), (d["cams0"], d["pts0"], d["single0"])):
        dst.copy_(src)
    d["cams"], d["pts"], d["single"] = views(state)
    state.copy_(state0)
    d["after"] = d["before"].clone()
    h_cams_np = np.zeros((n_kf, 6), np.float64)
    h_after = torch.empty((n_kf, 16), dtype=torch.float32).pin_memory()
    h_after.copy_(torch.from_numpy(poses_before))
    m_out = ctx.match_descriptors(d["q"], d["t"], nq, nt)
    t_out = ctx.triangulate_matches(d["kp1"], d["kp2"], m_out["mt"], m_out["mq"], m_out["cnt"], nq, d["poses"], pair["K"])
    last = {}

    def one_pass():
        ctx.match_descriptors(d["q"], d["t"], nq, nt, out=m_out)
        ctx.reproj_match(fv_a, mv_a, out=r_a)                   # match_key_frame
        ctx.reproj_match(fv_b, mv_b, out=r_b)                   # match_map
        ctx.triangulate_matches(d["kp1"], d["kp2"], m_out["mt"], m_out["mq"], m_out["cnt"], nq, d["poses"], pair["K"], out=t_out)
        last["window"] = rs.build_local_window(*lw_args)        # host, overlaps the kernels enqueued above
        state.copy_(state0)
        last["ba"] = ctx.bundle_adjust(d["cams"], window["cam_free"], d["pts"], d["optr"], d["ocam"], d["ouv"], window["K"])
        # poses are host-owned objects in the reference (Frame::set_pose): read back, unpack (f32), re-anchor
        ctx.ba_cameras(h_cams_np)
        rs.unpack_poses(h_cams_np, window["cam_free"], h_after.numpy())
        d["after"].copy_(h_after, non_blocking=True)
        ctx.reanchor_points(None, d["single_frame"], d["before"], d["after"], d["single"])

    def cpu_pass(O):
        mq, mt = O.match_descriptors(pair["desc2"], pair["desc1"])
        O.reproj_match(frame, mp_a)
        O.reproj_match(frame_b, mp_b)
        O.triangulate(pair["kp1"][mt], pair["kp2"][mq], pair["poses"], pair["K"])
        O.build_local_window(*lw_args)
        cams, pts, s = O.bundle_adjust(window["cams"], window["cam_free"], window["points"], window["obs_ptr"],
                                       window["obs_cam"], window["obs_uv"], window["K"])
        after = poses_before.copy()
        for c in free_idx:
            after[c] = O.unpack_pose(cams[c]).reshape(16)
        O.reanchor_points(None, single_frame, poses_before, after, single_pos)

    keep = (keep_fa, keep_ma, keep_fb, keep_mb)
    meta = dict(pair=pair, window=window, window_all=window_all, nq=nq, nt=nt, mp=mp, keep=keep, last=last,
                n_single=n_single, match_key_frame_points=int(mp_a["eligible"].sum()), match_map_points=int(elig_b.sum()))
    return one_pass, cpu_pass, meta


def bench_pass(e, args):
    ctx = e.ctx
    attach_comm(e)
    one_pass, cpu_pass, meta = build_pass(e)
    elapsed = timed(e, one_pass, args.steps, args.warmup)
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = e.world * args.steps / elapsed
    per_kernel = profiled(e, one_pass, args.steps)
    stats = ctx.ba_stats()
    window, nq, nt, mp = meta["window"], meta["nq"], meta["nt"], meta["mp"]
    work = algorithmic_work(window, int(np.sum(window["cam_free"])))
    pmc, pmc_file, pmc_fresh = pmc_traffic("pass")
    # K5: linearisation flops only on the rounds that relinearise; the Schur term once per speculative set
    k5_flops = (stats["fresh_rounds"] * work["lin_flops"] + stats["set_evaluations"] * work["schur_flops"]) / max(stats["rounds"], 1)
    sets_per_round = stats["set_evaluations"] / max(stats["rounds"], 1)
    rl = {}
    for name, bound, amount, note in (
            ("K5_ba_schur_mfma", "mfma", k5_flops, "avg over the solve's rounds: 500*M linearisation flops on relinearising rounds, "
                                                   "Schur flops per speculative set"),
            ("K7_ba_reduced_solve", "latency", work["solve_flops"],
             "single-workgroup block LDL^T: an 18-step dependency chain, neither the MFMA nor the HBM roof applies "
             "(DESIGN.md 4.2); the number to watch is us_per_block_step"),
            ("K8_ba_backsub_cost", "hbm", work["cost_bytes"] + 16 * work["M"] + 2 * 24 * work["P"], None),
            ("K78_ba_solve_backsub", "latency", sets_per_round * (work["solve_flops"] + work["backsub_flops"]),
             "K7 + K8 of a round in ONE launch (one K7 workgroup per speculative radius; K8's workgroups wait inside the launch "
             "for its hand-off): the 18-step block LDL^T chain sets the duration, neither the MFMA nor the HBM roof applies "
             "(DESIGN.md 4.2); flops = (n^3/3 + 410 M + 50 P) per evaluated radius; the number to watch is us_per_block_step"),
            ("K1_hamming_knn2", "valu-int", 16.0 * nq * nt, "8 xor + 8 popcount-accumulate per descriptor pair; HBM side: "
                                                            "%.0f KB per launch" % ((32.0 * (nq + nt) + 12.0 * nq) / 1e3)),
            ("K4_triangulate_dlt", "mfma", 2500.0 * nq, "fp64 VALU (no matrix work), priced against the fp64 peak"),
            ("K2_reproj_match", "hbm", 13.0 * len(mp["positions"]) + 40.0 * len(mp["obs_kf"]) + 48.0 * nq, None)):
        r = roofline_entry(name, bound, amount, per_kernel, pmc, note)
        if r:
            rl[name] = r
    for k7 in ("K7_ba_reduced_solve", "K78_ba_solve_backsub"):
        if k7 in rl:
            rl[k7]["us_per_block_step"] = rl[k7]["avg_launch_us"] / max(work["n"] / 6.0, 1.0)
            rl[k7]["workgroups_per_launch"] = "one per speculative radius (<= 3)" + (" + K8's" if k7.startswith("K78") else "")
    dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms"]) if per_kernel else None
    roofline = rl.get(dom)
    if roofline is not None:
        roofline = dict(roofline, timing="hip events per launch on the library stream, %d instrumented passes" % args.steps,
                        traffic_source=pmc_file, traffic_matches_kernel_sources=pmc_fresh)

    boundary = getattr(args, "boundary_result", None)     # measured in main() before this process touched the GPU
    cpu = None
    if e.rank == 0 and e.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cpu_pass, 1.0, "passes/s", "full passes of the same workload (all stages incl. both "
                           "reprojection matches, build_local_window, unpack, re-anchoring)", args.cpu_seconds)
    line = {
        "metric": METRIC, "value": value, "unit": "passes/s", "n_gpus": e.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8 Hamming (match), f64 (DLT SVD, BA), f32 (gates)", "data": "synthetic",
        "config": {"workload": "cfg2 pair (2000x2000 brute-force match + 2000-slot DLT triangulation) + two "
                               "reprojection-gated matches (match_key_frame, match_map; 2000 kp x 10k landmarks) + "
                               "Mapper::bundle_adjust on cfg3 (build_local_window, 20 KF x 10k landmarks x ~60k obs, "
                               "10 LM iterations, pose read-back, re-anchoring of 2000 single-observation points) per GPU",
                   "passes_per_step": e.world,
                   "ba_landmarks_total": int(len(meta["window_all"]["points"])),
                   "ba_obs_per_gpu": int(len(window["obs_cam"])),
                   "match_key_frame_points": meta["match_key_frame_points"], "match_map_points": meta["match_map_points"],
                   "parallelism": "landmark-sharded BA, RCCL all-reduce of the reduced camera system" if e.world > 1 else "single GPU"},
        "roofline": roofline,
        "cpu_baseline": cpu["one"] if cpu else None,
        "cpu_baseline_all_cores": cpu["all"] if cpu else None,
        "per_kernel_us": {k: round(v["avg_us"], 2) for k, v in sorted(per_kernel.items())},
        "per_kernel_launches_per_pass": {k: v["launches"] / max(args.steps, 1) for k, v in sorted(per_kernel.items())},
        "ba_summary": meta["last"].get("ba"), "ba_rounds": stats,
        "roofline_all": rl,
        "boundary": boundary,
    }
    finish(e, args, line)


# =============================================================================================== cfg2
def bench_cfg2(e, args):
    """configs[1]: one 1080p pair, 2000 ORB keypoints: brute-force match + DLT triangulation.  Every rank runs its own
    pair (replicas; no exchange step)."""
    ctx, synth = e.ctx, e.synth
    pair = synth.make_pair(2, seed_stream=e.rank)
    nq, nt = len(pair["desc2"]), len(pair["desc1"])
    dq, dt, dk1, dk2, dpo = [ctx.dev(pair[k]) for k in ("desc2", "desc1", "kp1", "kp2", "poses")]
    m_out = ctx.match_descriptors(dq, dt, nq, nt)
    t_out = ctx.triangulate_matches(dk1, dk2, m_out["mt"], m_out["mq"], m_out["cnt"], nq, dpo, pair["K"])

    def step():
        ctx.match_descriptors(dq, dt, nq, nt, out=m_out)
        ctx.triangulate_matches(dk1, dk2, m_out["mt"], m_out["mq"], m_out["cnt"], nq, dpo, pair["K"], out=t_out)

    def cpu_step(O):
        mq, mt = O.match_descriptors(pair["desc2"], pair["desc1"])
        O.triangulate(pair["kp1"][mt], pair["kp2"][mq], pair["poses"], pair["K"])

    elapsed = timed(e, step, args.steps, args.warmup)
    per_kernel = profiled(e, step, args.steps)
    pmc, pmc_file, pmc_fresh = pmc_traffic("cfg2")
    rl = {}
    for name, bound, amount in (("K1_hamming_knn2", "valu-int", 16.0 * nq * nt), ("K4_triangulate_dlt", "mfma", 2500.0 * nq)):
        r = roofline_entry(name, bound, amount, per_kernel, pmc)
        if r:
            rl[name] = r
    hb = roofline_entry("K1_hamming_knn2", "hbm", 32.0 * (nq + nt) + 12.0 * nq, per_kernel, pmc,
                        "the HBM view of K1 (BASELINE north_star asks for GB/s): intensity ~420 int-ops/B, so the VALU-int "
                        "roof in roofline_all is the binding one")
    cpu = None
    if e.rank == 0 and e.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cpu_step, 1.0, "pairs/s", "the same pair, match + triangulate", min(args.cpu_seconds, 10))
    finish(e, args, {
        "metric": "match+triangulate frame pairs/sec @ 1080p, 2000 ORB kpts (BASELINE configs[1])",
        "value": e.world * args.steps / elapsed, "unit": "pairs/s", "n_gpus": e.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8 Hamming, f64 DLT SVD, f32 gates", "data": "synthetic",
        "config": {"workload": "cfg2: 2000 x 2000 brute-force Hamming 2-NN + Lowe + DLT triangulation of the accepted pairs",
                   "parallelism": "independent pair per GPU (replicas only)"},
        "roofline": dict(hb, traffic_source=pmc_file, traffic_matches_kernel_sources=pmc_fresh) if hb else None,
        "cpu_baseline": cpu["one"] if cpu else None, "cpu_baseline_all_cores": cpu["all"] if cpu else None,
        "per_kernel_us": {k: round(v["avg_us"], 2) for k, v in sorted(per_kernel.items())}, "roofline_all": rl})


# =============================================================================================== cfg4
def bench_cfg4(e, args):
    """configs[3]: 64 key-frame pairs x 2000 keypoints; rank r takes pairs [r*64/N, (r+1)*64/N): no collective."""
    ctx, synth = e.ctx, e.synth
    B_total = 64
    if B_total % e.world:
        raise SystemExit("cfg4 shards 64 pairs: --gpus must divide 64")
    B = B_total // e.world
    bt = synth.make_pair_batch(B, first_stream=e.rank * B)
    n = bt["desc1"].shape[1]
    dq, dt, dk1, dk2, dpo = [ctx.dev(bt[k]) for k in ("desc2", "desc1", "kp1", "kp2", "poses")]
    m_out = ctx.match_descriptors(dq, dt, n, n, batch=B)
    t_out = ctx.triangulate_matches_batch(dk1, dk2, m_out["mt"], m_out["mq"], m_out["cnt"], dpo, bt["K"])

    def step():
        ctx.match_descriptors(dq, dt, n, n, batch=B, out=m_out)
        ctx.triangulate_matches_batch(dk1, dk2, m_out["mt"], m_out["mq"], m_out["cnt"], dpo, bt["K"], out=t_out)

    n_cpu = 4

    def cpu_step(O):
        for pr in bt["pairs"][:n_cpu]:
            mq, mt = O.match_descriptors(pr["desc2"], pr["desc1"])
            O.triangulate(pr["kp1"][mt], pr["kp2"][mq], pr["poses"], pr["K"])

    elapsed = timed(e, step, args.steps, args.warmup)
    per_kernel = profiled(e, step, args.steps)
    pmc, pmc_file, pmc_fresh = pmc_traffic("cfg4")
    rl = {}
    for name, bound, amount in (("K1_hamming_knn2", "valu-int", 16.0 * n * n * B), ("K4_triangulate_dlt", "mfma", 2500.0 * n * B),
                                ("K1b_merge_filter", "hbm", 16.0 * 4 * 4 * n * B)):
        r = roofline_entry(name, bound, amount, per_kernel, pmc)
        if r:
            rl[name] = r
    hb = roofline_entry("K1_hamming_knn2", "hbm", (32.0 * 2 * n + 12.0 * n) * B, per_kernel, pmc,
                        "HBM view of the batched K1 (algorithmic 152 KB per pair); the binding roof is VALU-int, see roofline_all")
    cpu = None
    if e.rank == 0 and e.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cpu_step, float(n_cpu), "pairs/s", "%d of the 64 pairs, match + triangulate" % n_cpu, min(args.cpu_seconds, 12))
    finish(e, args, {
        "metric": "match+triangulate frame pairs/sec @ batch of 64 KF-pairs x 2k kpts (BASELINE configs[3])",
        "value": B_total * args.steps / elapsed, "unit": "pairs/s", "n_gpus": e.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8 Hamming, f64 DLT SVD, f32 gates", "data": "synthetic",
        "config": {"workload": "cfg4: 64 pairs x (2000 x 2000 brute-force match + triangulation), one batched launch sequence",
                   "pairs_per_gpu": B, "parallelism": "pairs sharded over ranks, no collective"},
        "roofline": dict(hb, traffic_source=pmc_file, traffic_matches_kernel_sources=pmc_fresh) if hb else None,
        "cpu_baseline": cpu["one"] if cpu else None, "cpu_baseline_all_cores": cpu["all"] if cpu else None,
        "per_kernel_us": {k: round(v["avg_us"], 2) for k, v in sorted(per_kernel.items())}, "roofline_all": rl,
        "matches_per_pair": float(m_out["cnt"].float().mean().item()), "triangulated_per_pair": float(t_out["count"].float().mean().item())})


# ========================================================================================= cfg3 / cfg5
def bench_ba(e, args, cfg):
    """configs[2] (20 KF, 10k landmarks) / configs[4] (100 KF, 80k landmarks) as one 10-iteration BA.  N ranks:
    landmarks sharded N ways (cameras replicated), RCCL all-reduce of the reduced camera system per LM round."""
    ctx, synth, torch = e.ctx, e.synth, e.torch
    attach_comm(e)
    n_kf, n_pts = (20, 10000) if cfg == "cfg3" else (100, 80000)
    w_all = synth.make_ba_window(n_kf=n_kf, n_points=n_pts, config_id=3 if cfg == "cfg3" else 5)
    w = synth.shard_ba_by_landmark(w_all, e.world, e.rank) if e.world > 1 else w_all
    c0, p0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
    dc, dp = c0.clone(), p0.clone()
    dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
    last = {}

    def step():
        dc.copy_(c0)
        dp.copy_(p0)
        last["ba"] = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])

    def cpu_step(O):
        O.bundle_adjust(w_all["cams"], w_all["cam_free"], w_all["points"], w_all["obs_ptr"], w_all["obs_cam"], w_all["obs_uv"], w_all["K"])

    elapsed = timed(e, step, args.steps, args.warmup)
    per_kernel = profiled(e, step, args.steps)
    stats = ctx.ba_stats()
    work = algorithmic_work(w, int(np.sum(w["cam_free"])))
    pmc, pmc_file, pmc_fresh = pmc_traffic(cfg)
    k5_flops = (stats["fresh_rounds"] * work["lin_flops"] + stats["set_evaluations"] * work["schur_flops"]) / max(stats["rounds"], 1)
    rl = {}
    sets_per_round = stats["set_evaluations"] / max(stats["rounds"], 1)
    for name, bound, amount in (("K5_ba_schur_mfma", "mfma", k5_flops), ("K7_ba_reduced_solve", "latency", work["solve_flops"]),
                                ("K78_ba_solve_backsub", "latency", sets_per_round * (work["solve_flops"] + work["backsub_flops"])),
                                ("K7_ba_reduced_solve_blocked", "mfma", work["solve_flops"]),
                                ("K8_ba_backsub_cost", "hbm", work["cost_bytes"] + 16 * work["M"] + 2 * 24 * work["P"])):
        r = roofline_entry(name, bound, amount, per_kernel, pmc)
        if r:
            rl[name] = r
    dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms"]) if per_kernel else None
    roofline = rl.get(dom)
    if roofline is not None:
        roofline = dict(roofline, traffic_source=pmc_file, traffic_matches_kernel_sources=pmc_fresh)
    # throughput mode (reported beside `value`, never instead of it): B independent windows per call on the library's
    # lanes (rs_bundle_adjust_batch) — what a server holding several sessions on one GPU gets
    batch = None
    if e.world == 1 and cfg == "cfg3":
        batch = {}
        for mode, tag in ((0, "grid"), (1, "lanes")):
            ctx.set_int("ba_batch_mode", mode)
            for B in (8, 32, 128):
                if mode == 1 and B > 32:
                    continue
                clones = [(c0.clone(), p0.clone()) for _ in range(B)]
                probs = [(bc, w["cam_free"], bp, *dev, w["K"]) for bc, bp in clones]

                def bstep():
                    for bc, bp in clones:
                        bc.copy_(c0)
                        bp.copy_(p0)
                    torch.cuda.synchronize()
                    ctx.bundle_adjust_batch(probs)

                n_rep = max(3, args.steps // 5)
                dtb = timed(e, bstep, n_rep, 2)
                ent = dict(windows=B, solves_per_s=B * n_rep / dtb, ms_per_call=1e3 * dtb / n_rep,
                           speedup_vs_sequential=(B * n_rep / dtb) / (args.steps / elapsed))
                if mode == 0:
                    # what the batched grids do to the kernels: one launch serves B windows (same schedule per window as the
                    # single solve above, so the per-window work of a K5 launch is k5_flops)
                    pk = profiled(e, bstep, 2)
                    k5 = pk.get("K5_ba_schur_mfma")
                    if k5:
                        ent["per_kernel_us"] = {k: round(v["avg_us"], 1) for k, v in sorted(pk.items())}
                        ent["K5_tflops"] = B * k5_flops / (k5["avg_us"] * 1e-6) / 1e12
                        ent["K5_frac_of_f64_mfma_peak"] = ent["K5_tflops"] / 78.6
                batch["%s_B%d" % (tag, B)] = ent
        ctx.set_int("ba_batch_mode", 0)
        batch["note"] = ("B copies of the same cfg-3 window solved by one rs_bundle_adjust_batch call, incl. the 2B state-reset "
                         "copies.  grid: ONE launch sequence for all windows (blockIdx.z = window; the library's default); "
                         "lanes: 8 child contexts with their own streams, one host thread each")
    cpu = None
    if e.rank == 0 and e.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cpu_step, 1.0, "solves/s", "the same window, whole 10-iteration solve", args.cpu_seconds)
    name = "20-KF local BA solves/sec, 10k landmarks / ~60k obs (BASELINE configs[2])" if cfg == "cfg3" else \
           "100-KF global BA solves/sec, 80k landmarks / ~480k obs (BASELINE configs[4])"
    finish(e, args, {
        "metric": name, "value": args.steps / elapsed, "unit": "solves/s", "n_gpus": e.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d KF, %d landmarks, %d observations, 10 LM iterations" % (cfg, n_kf, n_pts, len(w_all["obs_cam"])),
                   "landmarks_per_gpu": int(len(w["points"])), "reduced_system_n": int(work["n"]),
                   "parallelism": "landmark-sharded, RCCL all-reduce of S | rhs | cost per LM round" if e.world > 1 else "single GPU"},
        "roofline": roofline, "cpu_baseline": cpu["one"] if cpu else None, "cpu_baseline_all_cores": cpu["all"] if cpu else None,
        "per_kernel_us": {k: round(v["avg_us"], 2) for k, v in sorted(per_kernel.items())},
        "per_kernel_launches_per_solve": {k: v["launches"] / max(args.steps, 1) for k, v in sorted(per_kernel.items())},
        "ba_summary": last.get("ba"), "ba_rounds": stats, "roofline_all": rl, "batch_throughput": batch})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="pass", choices=["pass", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", dest="boundary", action="store_false",
                    help="skip the end-to-end interface timings (tests/host_cpp/bench_boundary.bin) of --config pass")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="time budget of the CPU baseline leg")
    args = ap.parse_args()
    # The end-to-end interface timings run in a child process (tests/host_cpp/bench_boundary.bin).  It is started
    # BEFORE this process initialises the GPU: a process that holds a GPU context must not fork + exec on the box.
    if args.config == "pass" and args.boundary and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        args.boundary_result = boundary_timings()
    e = setup(args)
    if args.config == "pass":
        bench_pass(e, args)
    elif args.config == "cfg2":
        bench_cfg2(e, args)
    elif args.config == "cfg4":
        bench_cfg4(e, args)
    else:
        bench_ba(e, args, args.config)


if __name__ == "__main__":
    main()
