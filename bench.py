#!/usr/bin/env python3
"""bench.py — match + triangulate + local-BA passes/sec (BASELINE.json's metric).

One STEP = one pass of the hot path over one batch of synthetic input, inputs
resident in HBM before the timed region:
  (1) brute-force 2-NN Hamming + Lowe/threshold match, 2000 x 2000 ORB rows   (cfg 2, a4)
  (2) reprojection-gated match of the new frame against the window's landmarks (a2)
  (3) two-view DLT triangulation + gates of the accepted pairs                 (cfg 2, a6)
  (4) 10-iteration local bundle adjustment, 20 KF x 10k landmarks x ~60k obs   (cfg 3, a12)
N > 1 (one process per GPU, launched by torch.distributed.run): every rank runs
(1)-(3) on its own frame pair and owns a 10k-landmark shard of a 20-KF window
with N x 10k landmarks; the BA all-reduces the reduced camera system over
RCCL/xGMI each LM step.  value = N passes / step time (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(dominant kernel, HIP-event timed) and `cpu_baseline` (the oracle, rank 0, N=1).
"""
import argparse
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


def algorithmic_work(w, n_free):
    """Per-LM-iteration algorithmic work of the BA, unpadded and sparse."""
    k = np.diff(w["obs_ptr"]).astype(np.float64)
    M, P, C = float(k.sum()), float(len(k)), float(len(w["cams"]))
    n = 6.0 * n_free
    lin_flops = 500.0 * M
    schur_flops = float(np.sum(50 + 108 * k + 216 * k * (k + 1) / 2 + 72 * k))
    solve_flops = n ** 3 / 3
    lin_bytes = 16 * M + 2 * 24 * P + 2 * 48 * C + 8 * n * n
    cost_bytes = 16 * M + 24 * P + 48 * C
    return dict(M=M, P=P, C=C, lin_flops=lin_flops, schur_flops=schur_flops, solve_flops=solve_flops,
                lin_bytes=lin_bytes, cost_bytes=cost_bytes)


PMC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_v3_pmc.csv")
PMC_NAMES = {"K5_ba_schur_mfma": "ba_schur_mfma", "K7_ba_reduced_solve": "ba_reduced_solve_lds",
             "K8_ba_backsub_cost": "ba_backsub_cost4", "K1_hamming_knn2": "k1_hamming_knn2",
             "K1b_merge_filter": "k1_merge_filter", "K2_reproj_match": "k2_reproj_match",
             "K4_triangulate_dlt": "k4_triangulate"}


def pmc_traffic():
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs of this same benchmark; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  The
    counters cannot be collected from inside bench.py; the file is the record of the last collection."""
    out = {}
    try:
        with open(PMC_FILE) as fh:
            next(fh)
            for line in fh:
                f = line.strip().split(",")
                out[f[0]] = float(f[4])
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-passes", type=int, default=12)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("racing-slam_amd")
    rs, synth = pkg.rsgpu, pkg.synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = rs.Context(local_rank)
    if world > 1:
        uid = torch.zeros(128, dtype=torch.uint8, device=ctx.device)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(rs.Context.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        ctx.comm_init(bytes(uid.cpu().numpy().tobytes()), world, rank)

    # ---------------------------------------------------------------- inputs
    pair = synth.make_pair(2, seed_stream=rank)                         # cfg 2, this rank's frame pair
    window_all = synth.make_ba_window(n_kf=20, n_points=10000 * world)  # cfg 3 (x N landmarks)
    window = synth.shard_ba_by_landmark(window_all, world, rank) if world > 1 else window_all
    shard_for_match = dict(window_all)
    if world > 1:
        lo, hi = window["point_range"]
        o0, o1 = int(window_all["obs_ptr"][lo]), int(window_all["obs_ptr"][hi])
        shard_for_match.update(points=window_all["points"][lo:hi], points_true=window_all["points_true"][lo:hi],
                               obs_ptr=window["obs_ptr"], obs_cam=window["obs_cam"], obs_uv=window["obs_uv"])
        del o0, o1
    frame, mp = synth.make_match_scene(shard_for_match, n_keypoints=2000, kdtree_build=rs.kdtree_build)

    nq, nt = len(pair["desc2"]), len(pair["desc1"])
    d_q, d_t = ctx.dev(pair["desc2"]), ctx.dev(pair["desc1"])
    d_kp1, d_kp2 = ctx.dev(pair["kp1"]), ctx.dev(pair["kp2"])
    d_poses = ctx.dev(pair["poses"])
    fv, keep_f = ctx.make_frame_view(frame)
    mv, keep_m = ctx.make_map_view(mp)
    cams0, pts0 = ctx.dev(window["cams"]), ctx.dev(window["points"])
    d_cams, d_pts = cams0.clone(), pts0.clone()
    d_optr, d_ocam, d_ouv = ctx.dev(window["obs_ptr"]), ctx.dev(window["obs_cam"]), ctx.dev(window["obs_uv"])
    m_out = ctx.match_descriptors(d_q, d_t, nq, nt)
    r_out = ctx.reproj_match(fv, mv)
    t_out = ctx.triangulate_matches(d_kp1, d_kp2, m_out["mt"], m_out["mq"], m_out["cnt"], nq, d_poses, pair["K"])

    last = {}

    def one_pass():
        ctx.match_descriptors(d_q, d_t, nq, nt, out=m_out)
        ctx.reproj_match(fv, mv, out=r_out)
        ctx.triangulate_matches(d_kp1, d_kp2, m_out["mt"], m_out["mq"], m_out["cnt"], nq, d_poses, pair["K"], out=t_out)
        d_cams.copy_(cams0)
        d_pts.copy_(pts0)
        last["ba"] = ctx.bundle_adjust(d_cams, window["cam_free"], d_pts, d_optr, d_ocam, d_ouv, window["K"])

    for _ in range(args.warmup):
        one_pass()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = world * args.steps / elapsed

    # ------------------------------------------- per-kernel HIP-event timing
    ctx.prof_begin()
    for _ in range(args.steps):
        one_pass()
    prof = ctx.prof_end()
    per_kernel = {k: dict(launches=v[0], avg_us=1e3 * v[1] / max(v[0], 1), total_ms=v[1]) for k, v in prof.items()}
    dom = max(per_kernel, key=lambda k: per_kernel[k]["total_ms"]) if per_kernel else None
    n_free = int(np.sum(window["cam_free"]))
    work = algorithmic_work(window, n_free)
    roofline = None
    if dom is not None:
        avg_s = per_kernel[dom]["avg_us"] * 1e-6
        if dom.startswith("K5"):
            flops = work["lin_flops"] + work["schur_flops"]
            roofline = dict(kernel=dom, bound="mfma", achieved=flops / avg_s / 1e12, peak=FP64_PEAK_TFLOPS,
                            unit="TFLOP/s", traffic=None, algorithmic_flops_per_launch=flops)
        elif dom.startswith("K7"):
            flops = work["solve_flops"]
            roofline = dict(kernel=dom, bound="mfma", achieved=flops / avg_s / 1e12, peak=FP64_PEAK_TFLOPS,
                            unit="TFLOP/s", traffic=None, algorithmic_flops_per_launch=flops)
        elif dom.startswith("K8"):
            nbytes = work["cost_bytes"] + 16 * work["M"] + 2 * 24 * work["P"]
            roofline = dict(kernel=dom, bound="hbm", achieved=nbytes / avg_s / 1e9, peak=HBM_PEAK_GBS,
                            unit="GB/s", traffic=None, algorithmic_bytes_per_launch=nbytes)
        elif dom.startswith("K1_"):
            nbytes = 32.0 * (nq + nt) + 12.0 * nq
            roofline = dict(kernel=dom, bound="hbm", achieved=nbytes / avg_s / 1e9, peak=HBM_PEAK_GBS,
                            unit="GB/s", traffic=None, algorithmic_bytes_per_launch=nbytes)
        else:
            nbytes = work["lin_bytes"]
            roofline = dict(kernel=dom, bound="hbm", achieved=nbytes / avg_s / 1e9, peak=HBM_PEAK_GBS,
                            unit="GB/s", traffic=None, algorithmic_bytes_per_launch=nbytes)
        roofline["frac"] = roofline["achieved"] / roofline["peak"]
        roofline["avg_launch_us"] = per_kernel[dom]["avg_us"]
        roofline["timing"] = "hip events per launch on the library stream, %d instrumented passes" % args.steps
        pmc = pmc_traffic()
        if PMC_NAMES.get(dom) in pmc:
            roofline["traffic"] = pmc[PMC_NAMES[dom]]
            roofline["traffic_source"] = "profiles/round1_v3_pmc.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; 2*FETCH+WRITE bytes per launch)"
        if dom.startswith("K7"):
            roofline["note"] = ("latency-bound single-workgroup factorisation (18 dependent block steps); neither the "
                                "MFMA nor the HBM roof applies, see DESIGN.md 4.2")

    # ---------------- the same pass with the front end on a second stream (reported beside `value`)
    # The four stages of a pass are independent calls; a caller that keeps BA (Mapper) and the
    # matching / triangulation front end (Tracker) on separate streams overlaps them.  `value` above is
    # the strictly sequential single-stream figure; this one is informational.
    two_streams = None
    if world == 1:
        s2 = torch.cuda.Stream(device=ctx.device)
        ctx2 = rs.Context(local_rank)
        ctx2.use_stream(s2)

        def one_pass_2s():
            ctx2.match_descriptors(d_q, d_t, nq, nt, out=m_out)
            ctx2.reproj_match(fv, mv, out=r_out)
            ctx2.triangulate_matches(d_kp1, d_kp2, m_out["mt"], m_out["mq"], m_out["cnt"], nq, d_poses, pair["K"], out=t_out)
            d_cams.copy_(cams0)
            d_pts.copy_(pts0)
            ctx.bundle_adjust(d_cams, window["cam_free"], d_pts, d_optr, d_ocam, d_ouv, window["K"])

        for _ in range(max(args.warmup, 1)):
            one_pass_2s()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_pass_2s()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        two_streams = dict(value=args.steps / dt2, unit="passes/s", ms_per_step=1e3 * dt2 / max(args.steps, 1),
                           note="front end (match, reproj match, triangulate) on a second context/stream, BA on the first")
        ctx2.close()

    # ---------------- widened row, measured beside the metric, not part of `value`:
    # the body of Mapper::triangulate_tracks for one key frame (2000 tracks, <= 10 sightings each)
    tracks_stage = None
    if world == 1:
        tk = synth.make_tracks(n_tracks=2000)
        targs = (ctx.dev(tk["track_uv"]), ctx.dev(tk["sight_ptr"]), ctx.dev(tk["sight_pose"]), ctx.dev(tk["sight_uv"]),
                 ctx.dev(tk["poses"]), tk["kf_pose"], tk["K"])
        d_skip = ctx.dev(tk["skip"])
        tout = ctx.triangulate_tracks(*targs, d_skip=d_skip)
        torch.cuda.synchronize()
        ctx.prof_begin()
        t0 = time.perf_counter()
        for _ in range(max(args.steps, 1)):
            ctx.triangulate_tracks(*targs, d_skip=d_skip, out=tout)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / max(args.steps, 1)
        tprof = ctx.prof_end()
        tracks_stage = dict(n_tracks=2000, n_sightings=int(len(tk["sight_pose"])), calls_per_s=1.0 / wall,
                            us_per_call=1e6 * wall,
                            per_kernel_us={k: round(1e3 * v[1] / max(v[0], 1), 2) for k, v in tprof.items()},
                            accepted=int(tout["counts"].cpu()[0]))
        if not args.no_cpu_baseline:
            import pyoracle as O
            O.build()
            t0 = time.perf_counter()
            for _ in range(5):
                O.triangulate_tracks(tk["track_uv"], tk["sight_ptr"], tk["sight_pose"], tk["sight_uv"], tk["poses"],
                                     tk["kf_pose"], tk["K"], skip=tk["skip"])
            tracks_stage["cpu_us_per_call"] = 1e6 * (time.perf_counter() - t0) / 5

    # every hot kernel against its roof (same event timings)
    roofline_all = {}
    pmc_all = pmc_traffic()
    def add(name, bound, work_amount):
        if name not in per_kernel:
            return
        t = per_kernel[name]["avg_us"] * 1e-6
        peak = FP64_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS
        ach = work_amount / t / (1e12 if bound == "mfma" else 1e9)
        roofline_all[name] = dict(bound=bound, achieved=ach, peak=peak, unit="TFLOP/s" if bound == "mfma" else "GB/s",
                                  frac=ach / peak, avg_launch_us=per_kernel[name]["avg_us"],
                                  traffic=pmc_all.get(PMC_NAMES.get(name)))
    add("K5_ba_schur_mfma", "mfma", work["lin_flops"] + work["schur_flops"])
    add("K7_ba_reduced_solve", "mfma", work["solve_flops"])
    add("K8_ba_backsub_cost", "hbm", work["cost_bytes"] + 16 * work["M"] + 2 * 24 * work["P"])
    add("K1_hamming_knn2", "hbm", 32.0 * (nq + nt) + 12.0 * nq)
    add("K4_triangulate_dlt", "mfma", 2500.0 * nq)
    add("K2_reproj_match", "hbm", 13.0 * len(mp["positions"]) + 4.0 * len(mp["obs_kf"]) * 2 + 32.0 * len(mp["obs_kf"]) + 48.0 * nq)

    # ------------------------------------------------- CPU baseline (rank 0)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import pyoracle as O
        O.build()
        passes = max(1, args.cpu_passes)
        t0 = time.perf_counter()
        for _ in range(passes):
            mq, mt = O.match_descriptors(pair["desc2"], pair["desc1"])
            O.reproj_match(frame, mp)
            O.triangulate(pair["kp1"][mt], pair["kp2"][mq], pair["poses"], pair["K"])
            O.bundle_adjust(window["cams"], window["cam_free"], window["points"], window["obs_ptr"],
                            window["obs_cam"], window["obs_uv"], window["K"])
        dt = time.perf_counter() - t0
        cpu = dict(value=passes / dt, unit="passes/s", cores=1, kind="port",
                   sample="%d full passes of the same workload (oracle/liboracle.so, gcc -O2, 1 thread; "
                          "faithful restatement, not the reference binary)" % passes)

    # all-cores variant of the same CPU baseline (OpenMP build of the oracle); informational
    cpu_all = None
    if cpu is not None:
        import pyoracle as O
        ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncore = min(ncore, 16)          # a one-GPU box's CPU share is 16 cores, whatever the affinity mask says
        os.environ.setdefault("OMP_NUM_THREADS", str(ncore))
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        try:
            O.use_all_cores(True)
            passes = max(1, args.cpu_passes)
            t0 = time.perf_counter()
            for _ in range(passes):
                mq, mt = O.match_descriptors(pair["desc2"], pair["desc1"])
                O.reproj_match(frame, mp)
                O.triangulate(pair["kp1"][mt], pair["kp2"][mq], pair["poses"], pair["K"])
                O.bundle_adjust(window["cams"], window["cam_free"], window["points"], window["obs_ptr"],
                                window["obs_cam"], window["obs_uv"], window["K"])
            dt = time.perf_counter() - t0
            cpu_all = dict(value=passes / dt, unit="passes/s", cores=int(os.environ["OMP_NUM_THREADS"]), kind="port",
                           sample="%d full passes, OpenMP build of the oracle (queries, observations and landmark blocks "
                                  "in parallel; reproj match and triangulation serial)" % passes)
        finally:
            O.use_all_cores(False)

    if rank == 0:
        out = {
            "metric": "match+triangulate+local-BA passes/sec @ 2k kpts/frame, 20-KF x 10k-pt window",
            "value": value, "unit": "passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 Hamming (match), f64 (DLT SVD, BA), f32 (gates)", "data": "synthetic",
            "config": {"workload": "cfg2 pair (2000x2000 brute-force match + 2000-slot DLT triangulation) + "
                                   "reprojection-gated match (2000 kp x 10k landmarks) + cfg3 local BA "
                                   "(20 KF, 10k landmarks, ~60k obs, 10 LM iterations) per GPU",
                       "passes_per_step": world,
                       "ba_landmarks_total": int(len(window_all["points"])),
                       "ba_obs_per_gpu": int(len(window["obs_cam"])),
                       "parallelism": "landmark-sharded BA, RCCL all-reduce of the reduced camera system" if world > 1 else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "per_kernel_us": {k: round(v["avg_us"], 2) for k, v in sorted(per_kernel.items())},
            "per_kernel_launches_per_pass": {k: v["launches"] / max(args.steps, 1) for k, v in sorted(per_kernel.items())},
            "ba_summary": last.get("ba"),
            "roofline_all": roofline_all,
            "two_streams": two_streams,
            "tracks_stage": tracks_stage,
            "speedup_vs_cpu_baseline": (value / cpu["value"]) if cpu else None,
        }
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
