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
    (3b) Mapper::triangulate_tracks (src/Mapper.cpp:246-305): 2000 tracks, each triangulated from (first sighting,
        key frame) with gates (1.0, 4.0), reprojection check into <= 100 sightings, parallax rule, top-up to 100
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
METRIC = "match+triangulate+local-BA passes/sec @ 2k kpts/frame, 20-KF\u00d710k-pt window"      # BASELINE.json's string, character for character

# profile-scope name -> kernel name as rocprofv3 prints it (prefix match: template arguments and "void " are ignored)
PMC_NAMES = {"K5_ba_schur_mfma": "ba_schur_mfma", "K7_ba_reduced_solve": "ba_reduced_solve_lds", "K78_ba_solve_backsub": "ba_solve_backsub",
             "K8_ba_backsub_cost": "ba_backsub_cost4", "K1_hamming_knn2": "k1_hamming_knn2",
             "K1b_merge_filter": "k1_merge_filter", "K2_reproj_match": "k2_reproj_match_grouped",
             "K4_triangulate_dlt": "k4_triangulate", "K7_ba_reduced_solve_blocked": "ba_big_update",
             "K6_tracks": "k6_tracks", "K12_point_errors": "k12_point_errors", "K11_refine_pose": "ba_refine_pose"}
PMC_ROUNDS = ("round4", "round3", "round2")      # newest committed counter file first


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
    path = None
    for rnd in PMC_ROUNDS:
        cand = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{config}.csv")
        if os.path.exists(cand):
            path = cand
            break
    out, src_hash = {}, None
    if path is None:
        return {}, None, None
    try:
        with open(path) as fh:
            for line in fh:
                if line.startswith("#"):
                    if "kernel_source_hash=" in line:
                        src_hash = line.strip().split("kernel_source_hash=")[1].split()[0]
                    continue
                f = line.strip().rsplit(",", 4)          # kernel names may contain commas (template arguments)
                if len(f) == 5 and f[0] != "kernel":
                    name = f[0].strip().strip('"')
                    if name.startswith("void "):
                        name = name[5:]
                    name = name.split("<")[0].split("(")[0].strip()
                    try:
                        out.setdefault(name, float(f[4]))      # (rows are sorted by total traffic: the first instance of a template wins)
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
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(e.local_rank)
    if e.world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", e.local_rank))
    e.ndev = torch.cuda.device_count()
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


def comm_evidence(e, per_kernel):
    """What the exchange step ran over, as the communicator itself reports it (ncclCommCount), and the HIP-event time
    of the two collectives of a BA round on the library stream."""
    n, kind = e.ctx.comm_count()
    out = {"rccl_ranks": n if kind == 1 else None, "comm_kind": {0: "none", 1: "rccl", 2: "in-process group"}[kind],
           "comm_ranks": n}
    for k in ("C1_allreduce_system", "C2_allreduce_cost"):
        if k in per_kernel:
            out[k + "_us"] = round(per_kernel[k]["avg_us"], 2)
            out[k + "_per_step"] = per_kernel[k]["launches"]
    return out


def in_process_shards(e, w_all, n_shards, reps=5):
    """The landmark-sharded BA (the product's N > 1 path: rank-offset blocks, two all-reduces per round, redundant reduced
    solves) on ONE GPU: n contexts of this process, each with its own stream and host thread, joined by
    rs_comm_init_local (a deterministic on-device sum instead of RCCL).  Bounds the exchange-step overhead of the
    sharded form; it is NOT a scaling curve (the shards share one GPU)."""
    import threading
    torch, rs, synth = e.torch, e.rs, e.synth
    ctxs = [rs.Context(e.local_rank) for _ in range(n_shards)]
    streams = [torch.cuda.Stream(device=ctxs[0].device) for _ in range(n_shards)]
    for c, st in zip(ctxs, streams):
        c.use_stream(st)
    rs.Context.comm_init_local(ctxs)
    shards = [synth.shard_ba_by_landmark(w_all, n_shards, r) for r in range(n_shards)]
    res = [None] * n_shards
    bar = threading.Barrier(n_shards)

    def work(r):
        try:
            c, sh = ctxs[r], shards[r]
            with torch.cuda.stream(streams[r]):
                c0, p0 = c.dev(sh["cams"]), c.dev(sh["points"])
                dc, dp = c0.clone(), p0.clone()
                dev = [c.dev(sh[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
                times = []
                for _ in range(reps + 1):
                    dc.copy_(c0)
                    dp.copy_(p0)
                    streams[r].synchronize()
                    bar.wait()
                    t0 = time.perf_counter()
                    s = c.bundle_adjust(dc, sh["cam_free"], dp, *dev, sh["K"])
                    times.append(time.perf_counter() - t0)
                # one more, instrumented (HIP events around every launch: not among the timed repetitions): WHICH kernels
                # a sharded solve runs — the banded reduced solve and the fused K7 + K8 launch must be among them
                dc.copy_(c0)
                dp.copy_(p0)
                streams[r].synchronize()
                bar.wait()
                c.prof_begin()
                c.bundle_adjust(dc, sh["cam_free"], dp, *dev, sh["K"])
                prof = c.prof_end()
                res[r] = (min(times[1:]), s, {k: round(1e3 * v[1] / max(v[0], 1), 1) for k, v in prof.items()})
        except Exception as ex:      # noqa: BLE001
            res[r] = ex
            bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_shards)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    alive = any(t.is_alive() for t in th)
    for c in ctxs:
        if not alive:
            c.comm_destroy()
            c.close()
    if alive or any(isinstance(r, Exception) or r is None for r in res):
        return {"error": repr([r for r in res if isinstance(r, Exception)][:1]) if not alive else "a rank is stuck"}
    return {"ms_per_solve": 1e3 * max(r[0] for r in res), "iterations": res[0][1]["iterations"],
            "final_cost": res[0][1]["final_cost"], "per_kernel_us_rank0": res[0][2]}


def timed(e, fn, steps, warmup):
    """W untimed + exactly K timed steps between barrier + synchronize; MAX over ranks."""
    torch, dist = e.torch, e.dist

    def barrier():
        if e.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if hasattr(fn, "prepare"):
        fn.prepare(warmup + steps)                          # per-pass working copies, made before the clock starts
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
    if hasattr(fn, "prepare"):
        fn.prepare(steps)
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


def cpu_quota():
    """CPU share of this process: the cgroup quota (cpu.max / cfs_quota) where one is set, else the affinity mask."""
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    src = "affinity mask"
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                 # cgroup v2: "<quota|max> <period>"
            q, p = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
                src = "/sys/fs/cgroup/cpu.max"
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / p
                    src = "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"
        except (OSError, ValueError):
            pass
    n = n_aff if quota is None else max(1, min(n_aff, int(quota + 0.5)))
    return n, src, n_aff


def cpu_baseline(cpu_fn, units_per_call, unit, what, budget_s):
    """The oracle's sources rebuilt as a BASELINE (-O3 -march=native on this host; never the checker binary):
    1 thread, and all cores with OpenMP (mirrors Ceres' num_threads = hardware_concurrency, src/Optimization.cpp:122-132)."""
    import pyoracle as O
    model, ncpu, _ = O.cpu_model()
    ncore, quota_src, n_aff = cpu_quota()
    if quota_src == "affinity mask":
        ncore = max(1, min(ncore, 32))   # no quota to read: more OpenMP threads than a box's share only contend (the ladder below tries fewer)
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
        share = ncore
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
                          cpu_model=model, nproc=ncpu, cpu_share=share, cpu_share_source=quota_src, affinity_cpus=n_aff,
                          sample=what + "; the same build + OpenMP over queries / map points / correspondences / "
                                        "observations / landmark blocks; threads = best of {share, share/2, share/4}")
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


def pass_through_boundary(boundary, meta, with_cpu):
    """what ONE key-frame pass costs a caller that goes through the drop-in boundary with host objects (the
    reference's Tracker / Mapper unchanged), as the sum of the end-to-end medians of tests/host_cpp/bench_boundary.bin over the
    reference's call pattern (src/Tracker.cpp:232-248, src/Mapper.cpp:153-170,246-305,364-394) — three ways — beside `value`
    (device-resident arrays) and beside the CPU restatement's time for the same calls on this benchmark's inputs.
    The boundary scene is its own (20 KF / ~2000 keypoints / ~11 k map points, built by the C++ harness): the sums say what
    the call PATTERN costs, not what this pass's exact inputs cost."""
    if not boundary or "error" in boundary:
        return None
    us = lambda k: boundary[k]["median_us"]      # noqa: E731
    try:
        common = us("match_descriptors") + us("triangulate_points_frame_pair")
        n_tracks = boundary["triangulate_tracks_unchanged_mapper_loop"]["tracks"]
        four = common + us("match_key_frame") + us("match_map") + us("triangulate_tracks_unchanged_mapper_loop") + us("bundle_adjust")
        inc = common + us("match_key_frame") + us("match_map") + us("triangulate_tracks_inc") + us("bundle_adjust")
        res = (common + us("frame_create_resident") + us("match_key_frame_resident") + us("match_map_resident") +
               us("triangulate_tracks_inc") + us("bundle_adjust_resident"))
    except KeyError as ex:
        return {"error": "bench_boundary.bin lacks %s (rebuild with __graft_entry__.build())" % ex}
    out = {
        "four_replaced_translation_units": round(four / 1e3, 3),
        "plus_the_two_Mapper_inc_edits": round(inc / 1e3, 3),
        "plus_resident_map": round(res / 1e3, 3),
        "calls": "match_descriptors + triangulate_points (frame pair) + match_key_frame + match_map + the track stage "
                 "(%d x triangulate_points with ONE correspondence in the unchanged Mapper loop | one select_track_points call with the "
                 ".inc) + build_local_window + bundle_adjust; resident: rs_frame_create + rs_map_match x 2 + rs_map_bundle_adjust" % n_tracks,
        "track_stage_ms": {"unchanged_mapper_loop": round(us("triangulate_tracks_unchanged_mapper_loop") / 1e3, 3),
                           "inc": round(us("triangulate_tracks_inc") / 1e3, 3)},
        "note": "medians of tests/host_cpp/bench_boundary.bin (host objects in -> host results out, its own 11 k-point scene); "
                "`value` is the same call pattern on device-resident arrays",
    }
    if with_cpu:
        import pyoracle as O
        O.use_baseline("fast")
        try:
            calls = meta["cpu_stage_calls"](O)
            cpu_us = {}
            for k, fn in calls.items():
                m = cpu_measure(fn, 1.5 if k == "bundle_adjust" else 0.5, min_reps=3)
                cpu_us[k] = 1e6 * m["median_s"]
        finally:
            O.use_baseline(None)
        out["cpu_restatement_same_calls_ms"] = round(sum(cpu_us.values()) / 1e3, 3)
        out["cpu_restatement_per_call_us"] = {k: round(v, 1) for k, v in cpu_us.items()}
    return out


# =============================================================================================== pass
def build_pass(e, streams=1, graph=False):
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
    # working copies of what a pass modifies (cameras, points, re-anchored points).  Every pass needs them back at the initial
    # state, which is bookkeeping of the benchmark and not part of the path (a running system solves a NEW window each time):
    # timed() / profiled() ask for a ring of fresh copies before their clock starts (`prepare`), one per pass; only a pass
    # beyond the prepared ring resets its copy itself, by one device copy
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
    ring = [(state,) + views(state)]
    ring_pos = [0]

    def prepare(n):
        n = min(n, 2048)                                    # (0.5 GB of copies at most; longer runs fall back to the per-pass reset)
        while len(ring) < n:
            buf = torch.zeros_like(state0)
            ring.append((buf,) + views(buf))
        for buf, *_ in ring:
            buf.copy_(state0)
        torch.cuda.synchronize()
        ring_pos[0] = 0
    d["after"] = d["before"].clone()
    h_cams_np = np.zeros((n_kf, 6), np.float64)
    h_after = torch.empty((n_kf, 16), dtype=torch.float32).pin_memory()
    h_after.copy_(torch.from_numpy(poses_before))
    h_after_np = h_after.numpy()
    cam_free_u8 = np.ascontiguousarray(window["cam_free"], np.uint8)
    m_out = ctx.match_descriptors(d["q"], d["t"], nq, nt)
    t_out = ctx.triangulate_matches(d["kp1"], d["kp2"], m_out["mt"], m_out["mq"], m_out["cnt"], nq, d["poses"], pair["K"])
    # (3b) Mapper::triangulate_tracks (src/Mapper.cpp:246-305): 2000 tracks of the new key frame
    tk = synth.make_tracks(n_tracks=2000, config_id=6 + 100 * rank)
    tk_dev = (ctx.dev(tk["track_uv"]), ctx.dev(tk["sight_ptr"]), ctx.dev(tk["sight_pose"]), ctx.dev(tk["sight_uv"]), ctx.dev(tk["poses"]))
    tk_skip = ctx.dev(tk["skip"])
    # the rotation-dependent parallax requirement per first-sighting pose comes from the host's libm (what the shim passes,
    # src/Mapper.cpp:281-288): the accepted list is then bit-identical to the CPU path's
    tk_req = ctx.dev(rs.parallax_requirements(tk["poses"], tk["kf_pose"]))
    k_out = ctx.triangulate_tracks(*tk_dev, tk["kf_pose"], tk["K"], d_skip=tk_skip, d_required=tk_req)
    # stages timed BESIDE the pass (not in `value`): Mapper::cull_points' arithmetic over the window's points after the BA
    # (src/Mapper.cpp:396-431) and Tracker's per-frame refine_pose (src/Tracker.cpp:313) on the new frame's matched points
    cull_in = dict(positions=window["points"].astype(np.float32), obs_ptr=window["obs_ptr"], obs_pose=window["obs_cam"],
                   obs_uv=window["obs_uv"], poses=poses_before, K=window["K"])
    cull_dev = [ctx.dev(cull_in[k]) for k in ("positions", "obs_ptr", "obs_pose", "obs_uv", "poses")]
    c_out = ctx.point_errors(*cull_dev, window["K"])
    sel = np.flatnonzero(window["obs_cam"] == n_kf - 1)[:2000]
    ref_pts = window["points_true"][full_obs_pt[sel]] if world == 1 else window["points"][full_obs_pt[sel]]
    refine_in = dict(cam=window["cams"][n_kf - 1].copy(), points=np.ascontiguousarray(ref_pts, np.float64),
                     uv=np.ascontiguousarray(window["obs_uv"][sel]), K=window["K"])
    refine_dev = (ctx.dev(refine_in["points"]), ctx.dev(refine_in["uv"]))
    last = {}

    def cull_stage():
        ctx.point_errors(*cull_dev, window["K"], out=c_out)

    def refine_stage():
        last["refine"] = ctx.refine_pose(refine_in["cam"], *refine_dev, window["K"])

    # The four front-end chains of a pass — descriptor match -> triangulation, the two reprojection matches, the track
    # triangulation — share no data (inputs are the frame / the map as they were before the pass, every call has its own
    # outputs), and each is a handful of launches that fill a fraction of the chip.  `streams` > 1 issues them side by side:
    # the first chain on the library's stream, the others through contexts of their own on their own HIP streams, forked
    # from and joined back into the library's stream by events (rs_context_fork / rs_context_wait_for), so the pass stays self-contained (nothing of it starts
    # before the previous pass has ended, the bundle adjustment starts when all four chains are done).
    side = []
    if streams > 1:
        for _ in range(3):
            c = rs.Context(e.local_rank)
            st = torch.cuda.Stream(device=ctx.device)
            c.use_stream(st)
            side.append((c, st))
    last["side_contexts"] = side
    side_ctx = [c for c, _ in side]

    def front_end(serial):
        if serial or not side:
            ctx.match_descriptors(d["q"], d["t"], nq, nt, out=m_out)
            ctx.reproj_match(fv_a, mv_a, out=r_a)                   # match_key_frame
            ctx.reproj_match(fv_b, mv_b, out=r_b)                   # match_map
            ctx.triangulate_matches(d["kp1"], d["kp2"], m_out["mt"], m_out["mq"], m_out["cnt"], nq, d["poses"], pair["K"], out=t_out)
            ctx.triangulate_tracks(*tk_dev, tk["kf_pose"], tk["K"], d_skip=tk_skip, out=k_out, d_required=tk_req)      # Mapper::triangulate_tracks
            return
        ctx.fork(*side_ctx)                                     # nothing of this pass starts before the previous one has ended
        ctx.match_descriptors(d["q"], d["t"], nq, nt, out=m_out)
        ctx.triangulate_matches(d["kp1"], d["kp2"], m_out["mt"], m_out["mq"], m_out["cnt"], nq, d["poses"], pair["K"], out=t_out)
        side[0][0].triangulate_tracks(*tk_dev, tk["kf_pose"], tk["K"], d_skip=tk_skip, out=k_out, d_required=tk_req)
        side[1][0].reproj_match(fv_b, mv_b, out=r_b)            # match_map
        side[2][0].reproj_match(fv_a, mv_a, out=r_a)            # match_key_frame
        ctx.wait_for(*side_ctx)                                 # join: what follows on the library stream needs all four chains

    # ... and as ONE hipGraph: the library's front-end entry points only enqueue (kernel launches on the context's stream;
    # their workspaces are grow-only and warm after the first call), so the four chains with their fork / join edges are
    # stream-captured once and a pass replays them with a single graph launch — the host no longer issues ten launches
    # one by one while the GPU waits for them.
    fe_graph = None
    if side and graph:
        for _ in range(3):
            front_end(False)
        torch.cuda.synchronize()
        cap = torch.cuda.Stream(device=ctx.device)
        fe_graph = torch.cuda.CUDAGraph()
        ctx.use_stream(cap)                                     # (the legacy default stream cannot be captured)
        try:
            with torch.cuda.graph(fe_graph, stream=cap, capture_error_mode="relaxed"):
                front_end(False)
        finally:
            ctx.use_stream(None)
        last["front_end_graph"] = fe_graph

    def one_pass(serial=False):
        if fe_graph is not None and not serial:
            fe_graph.replay()
        else:
            front_end(serial)
        last["window"] = rs.build_local_window(*lw_args)        # host, overlaps the kernels enqueued above
        if ring_pos[0] < len(ring):
            buf, d["cams"], d["pts"], d["single"] = ring[ring_pos[0]]
            ring_pos[0] += 1
        else:
            buf, d["cams"], d["pts"], d["single"] = ring[0]
            buf.copy_(state0)
        last["ba"] = ctx.bundle_adjust(d["cams"], window["cam_free"], d["pts"], d["optr"], d["ocam"], d["ouv"], window["K"])
        # poses are host-owned objects in the reference (Frame::set_pose): read back, unpack (f32), re-anchor
        ctx.ba_cameras(h_cams_np)
        rs.unpack_poses(h_cams_np, cam_free_u8, h_after_np)
        ctx.reanchor_points_host_poses(None, d["single_frame"], poses_before, h_after_np, d["single"])   # (poses as kernel arguments)

    def one_pass_serial():
        one_pass(True)

    one_pass.prepare = prepare
    one_pass_serial.prepare = prepare
    # set-up, not warm-up: the first bundle adjustment of a process loads its kernels' code objects and allocates the solver's
    # workspace (milliseconds); every other stage has run once above.  --warmup W passes still follow in timed().
    for _ in range(2):
        one_pass_serial()
    if side:
        one_pass()
    torch.cuda.synchronize()

    def cpu_pass(O, keep_results=None):
        mq, mt = O.match_descriptors(pair["desc2"], pair["desc1"])
        ra = O.reproj_match(frame, mp_a)
        rb = O.reproj_match(frame_b, mp_b)
        tri = O.triangulate(pair["kp1"][mt], pair["kp2"][mq], pair["poses"], pair["K"])
        trk = O.triangulate_tracks(tk["track_uv"], tk["sight_ptr"], tk["sight_pose"], tk["sight_uv"], tk["poses"], tk["kf_pose"],
                                   tk["K"], skip=tk["skip"])
        lw = O.build_local_window(*lw_args)
        cams, pts, s = O.bundle_adjust(window["cams"], window["cam_free"], window["points"], window["obs_ptr"],
                                       window["obs_cam"], window["obs_uv"], window["K"])
        after = poses_before.copy()
        for c in free_idx:
            after[c] = O.unpack_pose(cams[c]).reshape(16)
        single = O.reanchor_points(None, single_frame, poses_before, after, single_pos)
        if keep_results is not None:
            keep_results.update(mq=mq, mt=mt, ra=ra, rb=rb, tri=tri, trk=trk, lw=lw, cams=cams, pts=pts, ba=s, after=after, single=single)

    def cpu_stage_calls(O):
        """The pass's interface calls one by one (the same oracle functions as cpu_pass), for per-call CPU times."""
        mq, mt = O.match_descriptors(pair["desc2"], pair["desc1"])
        return {
            "match_descriptors": lambda: O.match_descriptors(pair["desc2"], pair["desc1"]),
            "triangulate_points_frame_pair": lambda: O.triangulate(pair["kp1"][mt], pair["kp2"][mq], pair["poses"], pair["K"]),
            "match_key_frame": lambda: O.reproj_match(frame, mp_a),
            "match_map": lambda: O.reproj_match(frame_b, mp_b),
            "triangulate_tracks": lambda: O.triangulate_tracks(tk["track_uv"], tk["sight_ptr"], tk["sight_pose"], tk["sight_uv"], tk["poses"],
                                                               tk["kf_pose"], tk["K"], skip=tk["skip"]),
            "build_local_window": lambda: O.build_local_window(*lw_args),
            "bundle_adjust": lambda: O.bundle_adjust(window["cams"], window["cam_free"], window["points"], window["obs_ptr"],
                                                     window["obs_cam"], window["obs_uv"], window["K"]),
        }

    def cpu_cull(O):
        return O.point_errors(cull_in["positions"], cull_in["obs_ptr"], cull_in["obs_pose"], cull_in["obs_uv"], cull_in["poses"], cull_in["K"])

    def cpu_refine(O):
        return O.refine_pose(refine_in["cam"], refine_in["points"], refine_in["uv"], refine_in["K"])

    def gpu_results():
        """What the last GPU pass left behind (host copies), in the same keys as cpu_pass's keep_results."""
        torch.cuda.synchronize()
        ctx.synchronize()
        g = lambda t: t.detach().cpu().numpy()      # noqa: E731
        cnt = int(g(m_out["cnt"])[0])
        na, nb_ = int(g(r_a["count"])[0]), int(g(r_b["count"])[0])
        nt_ = int(g(t_out["count"])[0])
        kc = g(k_out["counts"])
        return dict(mq=g(m_out["mq"])[0, :cnt], mt=g(m_out["mt"])[0, :cnt],
                    ra=dict(match_kp=g(r_a["match_kp"])[:na], match_point=g(r_a["match_point"])[:na]),
                    rb=dict(match_kp=g(r_b["match_kp"])[:nb_], match_point=g(r_b["match_point"])[:nb_]),
                    tri=dict(keep=g(t_out["keep"])[:cnt], xyz=g(t_out["xyz"])[:cnt], out_index=g(t_out["out_index"])[:nt_]),
                    trk=dict(status=g(k_out["status"])[:2000], xyz=g(k_out["xyz"])[:2000], accepted=g(k_out["accepted"])[:kc[0]],
                             inconsistent=g(k_out["inconsistent"])[:kc[2]], parallax_cos=g(k_out["parallax_cos"])[:2000],
                             required_cos=g(k_out["required_cos"])[:2000]),
                    lw=last.get("window"), cams=g(d["cams"]), pts=g(d["pts"]), ba=last.get("ba"), after=h_after.numpy().copy(),
                    single=g(d["single"]), cull=dict(mean_err=g(c_out["mean_err"]), cull=g(c_out["cull"])),
                    refine=last.get("refine"))

    keep = (keep_fa, keep_ma, keep_fb, keep_mb)
    meta = dict(pair=pair, window=window, window_all=window_all, nq=nq, nt=nt, mp=mp, keep=keep, last=last,
                n_single=n_single, match_key_frame_points=int(mp_a["eligible"].sum()), match_map_points=int(elig_b.sum()),
                frame_a=frame, mp_a=mp_a, frame_b=frame_b, mp_b=mp_b, tracks=tk, cull_in=cull_in, refine_in=refine_in,
                cull_stage=cull_stage, refine_stage=refine_stage, cpu_cull=cpu_cull, cpu_refine=cpu_refine, cpu_stage_calls=cpu_stage_calls,
                gpu_results=gpu_results, n_track_sightings=int(tk["sight_ptr"][-1]), one_pass_serial=one_pass_serial, front_end=front_end)
    return one_pass, cpu_pass, meta


def check_pass_parity(meta, cpu_pass, O):
    """The CPU leg computes the oracle's answer on the very inputs of the timed pass: compare it with what the GPU's LAST
    pass left behind.  Integer / index outputs, keep flags, triangulated and re-anchored positions: bit for bit (the
    contract of tests/test_gpu_parity.py); the BA: identical LM schedule, cost to 1e-7, poses / points to 1e-6.
    Returns (ok, report)."""
    ref = {}
    cpu_pass(O, ref)
    got = meta["gpu_results"]()
    rep, bad = {}, []

    def eq(name, a, b):
        ok = bool(np.array_equal(np.asarray(a), np.asarray(b)))
        rep[name] = ok
        if not ok:
            bad.append(name)

    def bits(name, a, b):
        a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        eq(name, a.view(np.uint32), b.view(np.uint32))

    eq("match_descriptors.query", got["mq"], ref["mq"])
    eq("match_descriptors.train", got["mt"], ref["mt"])
    for k, tag in (("ra", "match_key_frame"), ("rb", "match_map")):
        eq(tag + ".keypoints", got[k]["match_kp"], ref[k]["match_kp"])
        eq(tag + ".points", got[k]["match_point"], ref[k]["match_point"])
    eq("triangulate.keep", got["tri"]["keep"], ref["tri"]["keep"])
    bits("triangulate.xyz", got["tri"]["xyz"], ref["tri"]["xyz"])
    eq("triangulate.out_index", got["tri"]["out_index"], ref["tri"]["out_index"])
    eq("tracks.status", got["trk"]["status"], ref["trk"]["status"])
    bits("tracks.xyz", got["trk"]["xyz"], ref["trk"]["xyz"])
    eq("tracks.inconsistent", got["trk"]["inconsistent"], ref["trk"]["inconsistent"])
    bits("tracks.required_cos", got["trk"]["required_cos"], ref["trk"]["required_cos"])      # host-libm table: bit-exact
    eq("tracks.accepted", got["trk"]["accepted"], ref["trk"]["accepted"])
    eq("build_local_window.frames", got["lw"][0], ref["lw"][0])
    eq("build_local_window.optimize", got["lw"][1], ref["lw"][1])
    s, r = got["ba"], ref["ba"]
    sched = tuple(s[k] for k in ("iterations", "successful_steps", "termination", "usable")) == \
        tuple(r[k] for k in ("iterations", "successful_steps", "termination", "usable"))
    rep["bundle_adjust.schedule"] = bool(sched)
    rep["bundle_adjust.final_cost_rel_err"] = abs(s["final_cost"] - r["final_cost"]) / max(abs(r["final_cost"]), 1e-300)
    rep["bundle_adjust.cameras_max_abs_err"] = float(np.abs(got["cams"] - ref["cams"]).max())
    rep["bundle_adjust.points_max_abs_err"] = float(np.abs(got["pts"] - ref["pts"]).max())
    if not sched or rep["bundle_adjust.final_cost_rel_err"] > 1e-7 or \
            not np.allclose(got["cams"], ref["cams"], rtol=1e-6, atol=1e-8) or not np.allclose(got["pts"], ref["pts"], rtol=1e-6, atol=1e-7):
        bad.append("bundle_adjust")
    # the f32 poses come from f64 cameras that agree to ~1e-9: an f32 rounding boundary may flip a last bit
    rep["unpack_poses.max_abs_err"] = float(np.abs(got["after"] - ref["after"]).max())
    if rep["unpack_poses.max_abs_err"] > 1e-5:
        bad.append("unpack_poses")
    rep["reanchor.max_abs_err"] = float(np.abs(got["single"] - ref["single"]).max())
    if rep["reanchor.max_abs_err"] > 1e-3:
        bad.append("reanchor")
    return not bad, dict(rep, mismatches=bad)


def bench_pass(e, args):
    ctx = e.ctx
    attach_comm(e)
    # `value` is the pass as a serial caller issues it: every launch on ONE stream (--streams 1, the default).  The same pass
    # with its four front-end chains side by side on streams of their own is timed beside it (or becomes `value` with
    # --streams 4): that form assumes a caller whose chains are data-independent — true of this benchmark's inputs; in the
    # reference's frame flow Mapper::triangulate_tracks consumes the matches of the same frame (src/Mapper.cpp:246-305).
    overlapped, cpu_pass, meta = build_pass(e, 4, graph=args.graph)
    serial = meta["one_pass_serial"]
    one_pass = overlapped if args.streams > 1 else serial
    other = serial if args.streams > 1 else overlapped
    elapsed = timed(e, one_pass, args.steps, args.warmup)
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = e.world * args.steps / elapsed
    retries_before = ctx.ba_stats()["handoff_retries"]
    other_ms = 1e3 * timed(e, other, args.steps, args.warmup) / max(args.steps, 1)
    # (front-end chains on forked streams can hold compute units the fused K7 + K8 launch counts on; a lost hand-off
    # is re-run as separate launches — correct, but a 4 ms spike: count them per form)
    retries_other = ctx.ba_stats()["handoff_retries"] - retries_before
    per_kernel = profiled(e, serial, args.steps)     # (HIP events of one context on one stream)
    one_pass()                                       # the parity check below looks at what the LAST pass left behind: make it `value`'s form
    stats = ctx.ba_stats()
    # in-run parity: the oracle on the very inputs of the timed pass against what the GPU's last pass left behind
    parity = None
    if e.rank == 0 and e.world == 1 and not args.no_cpu_baseline:
        import pyoracle as O
        ok, parity = check_pass_parity(meta, cpu_pass, O)
        if not ok:
            raise SystemExit("bench.py: the GPU pass differs from the oracle on the benchmark's own inputs: %s" % json.dumps(parity))
    # stages beside the pass (Mapper::cull_points' arithmetic, Tracker's refine_pose): wall time per call + kernel time
    stages = {}
    for tag, fn in (("cull_stage", meta["cull_stage"]), ("refine_stage", meta["refine_stage"])):
        dt = timed(e, fn, args.steps, args.warmup)
        pk = profiled(e, fn, args.steps)
        stages[tag] = dict(us_per_call=1e6 * dt / max(args.steps, 1), per_kernel_us={k: round(v["avg_us"], 2) for k, v in sorted(pk.items())})
        per_kernel_side = pk
        stages[tag]["_pk"] = per_kernel_side
    empty_launch_us = ctx.empty_launch_us()
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
            ("K6_tracks", "mfma", 2500.0 * 2000 + 60.0 * meta["n_track_sightings"],
             "Mapper::triangulate_tracks body: one DLT per track (2.5 kflop fp64 VALU) + an f32 reprojection per sighting; "
             "a dependent fp64 chain per lane, priced against the fp64 peak"),
            ("K2_reproj_match", "hbm", 13.0 * len(mp["positions"]) + 40.0 * len(mp["obs_kf"]) + 48.0 * nq, None)):
        r = roofline_entry(name, bound, amount, per_kernel, pmc, note)
        if r:
            rl[name] = r
    # roofline entries of the side stages: K12 streams the observation CSR (16 B per observation + 12 B per point in, 5 B out);
    # K11 is one workgroup running the whole LM loop (latency)
    M_w, P_w = float(len(window["obs_cam"])), float(len(window["points"]))
    r = roofline_entry("K12_point_errors", "hbm", 16.0 * M_w + 17.0 * P_w, stages["cull_stage"]["_pk"], pmc)
    if r:
        rl["K12_point_errors"] = r
    r = roofline_entry("K11_refine_pose", "latency", 2.0 * 400.0 * len(meta["refine_in"]["uv"]) * 4, stages["refine_stage"]["_pk"], pmc,
                       "one workgroup, the whole LM loop in one launch: flops = (linearise + cost pass) x iterations, ~400 per observation and pass")
    if r:
        rl["K11_refine_pose"] = r
    for st in stages.values():
        st.pop("_pk", None)
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
                           "reprojection matches, the 2000-track triangulation, build_local_window, unpack, re-anchoring)", args.cpu_seconds)
        import pyoracle as O
        O.use_baseline("fast")
        try:
            for tag, fn in (("cull_stage", meta["cpu_cull"]), ("refine_stage", meta["cpu_refine"])):
                m = cpu_measure(lambda: fn(O), 2.0, min_reps=10)
                stages[tag]["cpu_us_per_call_1_thread"] = 1e6 * m["median_s"]
        finally:
            O.use_baseline(None)
    through = pass_through_boundary(boundary, meta, cpu is not None)
    line = {
        "metric": METRIC, "value": value, "unit": "passes/s", "n_gpus": e.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8 Hamming (match), f64 (DLT SVD, BA), f32 (gates)", "data": "synthetic",
        "pass_through_boundary_ms": through,
        "config": {"workload": "cfg2 pair (2000x2000 brute-force match + 2000-slot DLT triangulation) + two "
                               "reprojection-gated matches (match_key_frame, match_map; 2000 kp x 10k landmarks) + "
                               "Mapper::triangulate_tracks (2000 tracks, per-track DLT + sighting checks + quota) + "
                               "Mapper::bundle_adjust on cfg3 (build_local_window, 20 KF x 10k landmarks x ~60k obs, "
                               "10 LM iterations, pose read-back, re-anchoring of 2000 single-observation points) per GPU",
                   "streams": ("the four independent front-end chains (match -> triangulate | match_key_frame | match_map | "
                               "triangulate_tracks) side by side on 4 HIP streams, forked from / joined into the library stream "
                               "by events; the bundle adjustment starts when all four are done" +
                               ("; captured once, replayed as one hipGraph launch per pass" if args.graph else "")) if args.streams > 1 else "one stream",
                   "state_reset": "every pass starts from a fresh copy of the window (cameras, points, re-anchored points) made before the "
                                  "clock starts — a ring of one copy per pass; the reset is bookkeeping of the benchmark, not part of the path",
                   "passes_per_step": e.world,
                   "ba_landmarks_total": int(len(meta["window_all"]["points"])),
                   "ba_obs_per_gpu": int(len(window["obs_cam"])),
                   "match_key_frame_points": meta["match_key_frame_points"], "match_map_points": meta["match_map_points"],
                   "parallelism": "landmark-sharded BA, RCCL all-reduce of the reduced camera system" if e.world > 1 else "single GPU"},
        "roofline": roofline,
        "cpu_baseline": cpu["one"] if cpu else None,
        "cpu_baseline_all_cores": cpu["all"] if cpu else None,
        ("ms_per_step_one_stream" if args.streams > 1 else "ms_per_step_front_end_on_4_streams"): other_ms,
        "front_end_streams_note": "the same pass with match->triangulate | match_key_frame | match_map | triangulate_tracks side by side on four "
                                  "HIP streams (rs_context_wait_for fork / join): valid for a caller whose four chains are data-independent, "
                                  "as this benchmark's inputs are; not `value` by default because in the reference's frame flow "
                                  "triangulate_tracks consumes the same frame's matches",
        "handoff_retries": {"value_form": int(stats["handoff_retries"] - retries_other),
                            ("one_stream_form" if args.streams > 1 else "front_end_on_4_streams_form"): int(retries_other),
                            "note": "solves re-run as separate launches because a K8 workgroup of the fused K7 + K8 launch timed out waiting "
                                    "for its hand-off word (rs_ba_get_stats [4]); 0 expected on an otherwise idle GPU"},
        "per_kernel_us": {k: round(v["avg_us"], 2) for k, v in sorted(per_kernel.items())},
        "per_kernel_launches_per_pass": {k: v["launches"] / max(args.steps, 1) for k, v in sorted(per_kernel.items())},
        "ba_summary": meta["last"].get("ba"), "ba_rounds": stats,
        "roofline_all": rl,
        "stages_beside_the_pass": dict(stages, note="not in `value`: Mapper::cull_points' arithmetic (rs_point_errors over the window's "
                                       "10k points / ~60k observations) and Tracker's per-frame refine_pose (2000 observations); "
                                       "wall time per call incl. the host hand-off, kernel time, CPU restatement 1 thread"),
        "parity_checked_in_run": bool(parity is not None), "parity": parity,
        "empty_launch_us": round(empty_launch_us, 3),
        "boundary": boundary,
    }
    if e.world > 1:
        line.update(comm_evidence(e, per_kernel))
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

    # every solve starts from a fresh copy of the window, made before the clock starts (as the pass does: the reset is
    # bookkeeping of the benchmark); a solve beyond the prepared ring resets its copy itself
    ring, ring_pos = [(dc, dp)], [0]

    def prepare(n):
        n = min(n, 512 if cfg == "cfg3" else 64)
        while len(ring) < n:
            ring.append((c0.clone(), p0.clone()))
        for a, b_ in ring:
            a.copy_(c0)
            b_.copy_(p0)
        torch.cuda.synchronize()
        ring_pos[0] = 0

    def step():
        if ring_pos[0] < len(ring):
            a, b_ = ring[ring_pos[0]]
            ring_pos[0] += 1
        else:
            a, b_ = ring[0]
            a.copy_(c0)
            b_.copy_(p0)
        last["ba"] = ctx.bundle_adjust(a, w["cam_free"], b_, *dev, w["K"])

    step.prepare = prepare
    step()                                                   # set-up: code objects, workspace (not a warm-up pass)
    torch.cuda.synchronize()

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
    want = getattr(args, "in_process_shards", 0)
    if e.world == 1 and cfg == "cfg3" and not want:
        batch = {}
        for mode, tag in ((0, "grid"), (1, "lanes")):
            ctx.set_int("ba_batch_mode", mode)
            for B in (8, 32, 128):
                if mode == 1 and B > 32:
                    continue
                # every call solves B fresh copies of the window, made before the clock starts (a ring of problem sets)
                sets, set_pos = [], [0]

                def new_set():
                    clones = [(c0.clone(), p0.clone()) for _ in range(B)]
                    return clones, [(bc, w["cam_free"], bp, *dev, w["K"]) for bc, bp in clones]

                def bprepare(n, sets=sets, set_pos=set_pos, new_set=new_set):
                    while len(sets) < min(n, 16):
                        sets.append(new_set())
                    for clones, _ in sets:
                        for bc, bp in clones:
                            bc.copy_(c0)
                            bp.copy_(p0)
                    torch.cuda.synchronize()
                    set_pos[0] = 0

                def bstep(sets=sets, set_pos=set_pos):
                    if set_pos[0] < len(sets):
                        clones, probs = sets[set_pos[0]]
                        set_pos[0] += 1
                    else:                                   # beyond the prepared ring: reset one set inside the call
                        clones, probs = sets[0]
                        for bc, bp in clones:
                            bc.copy_(c0)
                            bp.copy_(p0)
                        torch.cuda.synchronize()
                    ctx.bundle_adjust_batch(probs)

                bstep.prepare = bprepare
                bprepare(1)

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
        batch["note"] = ("B fresh copies of the same cfg-3 window solved by one rs_bundle_adjust_batch call (the copies are made before "
                         "the clock starts).  grid: ONE launch sequence for all windows (blockIdx.z = window; the library's default); "
                         "lanes: 8 child contexts with their own streams, one host thread each")
    # the sharded form of the solve on this ONE GPU (in-process group): bounds what the exchange step costs
    shards = None
    if e.world == 1 and (want or not args.no_shard_rehearsal):
        shards = {"unsharded_ms_per_solve": 1e3 * elapsed / max(args.steps, 1)}
        for n_sh in ((want,) if want else (2, 4, 8)):
            shards["shards_%d" % n_sh] = in_process_shards(e, w_all, n_sh, reps=3 if cfg == "cfg5" else 5)
        shards["note"] = ("NOT a scaling curve: n landmark shards of the same window solved side by side on ONE GPU by n contexts of "
                          "this process (rs_comm_init_local; own stream and host thread each; two on-device sum all-reduces per LM "
                          "round instead of RCCL).  It executes the product's N > 1 code path and bounds the overhead of the exchange "
                          "steps; the shards share the GPU's CUs, so no speed-up is expected")
    cpu = None
    if e.rank == 0 and e.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cpu_step, 1.0, "solves/s", "the same window, whole 10-iteration solve", args.cpu_seconds)
    name = "20-KF local BA solves/sec, 10k landmarks / ~60k obs (BASELINE configs[2])" if cfg == "cfg3" else \
           "100-KF global BA solves/sec, 80k landmarks / ~480k obs (BASELINE configs[4])"
    extra = comm_evidence(e, per_kernel) if e.world > 1 else {}
    if want and shards and "ms_per_solve" in shards.get("shards_%d" % want, {}):
        extra["requested_gpus"] = want
        extra["note"] = ("--gpus %d on a box with %d GPU(s): the %d landmark shards ran in ONE process on ONE GPU (in_process_shards); "
                         "`value` is the unsharded single-GPU figure, n_gpus = 1; no multi-GPU scaling was measured" % (want, e.ndev, want))
    finish(e, args, {
        **extra,
        "empty_launch_us": round(ctx.empty_launch_us(), 3),
        "metric": name, "value": args.steps / elapsed, "unit": "solves/s", "n_gpus": e.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d KF, %d landmarks, %d observations, 10 LM iterations" % (cfg, n_kf, n_pts, len(w_all["obs_cam"])),
                   "landmarks_per_gpu": int(len(w["points"])), "reduced_system_n": int(work["n"]),
                   "parallelism": "landmark-sharded, RCCL all-reduce of S | rhs | cost per LM round" if e.world > 1 else "single GPU"},
        "roofline": roofline, "cpu_baseline": cpu["one"] if cpu else None, "cpu_baseline_all_cores": cpu["all"] if cpu else None,
        "per_kernel_us": {k: round(v["avg_us"], 2) for k, v in sorted(per_kernel.items())},
        "per_kernel_launches_per_solve": {k: v["launches"] / max(args.steps, 1) for k, v in sorted(per_kernel.items())},
        "ba_summary": last.get("ba"), "ba_rounds": stats, "roofline_all": rl, "batch_throughput": batch,
        "in_process_shards": shards})


def visible_gpu_count():
    """GPUs this process would see, WITHOUT a HIP call (a process that has touched the GPU must not fork + exec a launcher on
    the box; torch.cuda.device_count() stays clear of HIP only while amdsmi imports).  KFD topology nodes with SIMDs are the
    GPUs; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow the count."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in sorted(os.listdir(base)):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
    except OSError:
        n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids)) if n else len(ids)
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="pass", choices=["pass", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--streams", type=int, default=1, choices=[1, 4],
                    help="pass: 1 = every launch on one stream (default), 4 = the four front-end chains side by side on HIP streams of their own")
    ap.add_argument("--graph", action="store_true", help="pass, overlapped form: replay the front-end chains as one captured hipGraph instead of issuing them call by call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", dest="boundary", action="store_false",
                    help="skip the end-to-end interface timings (tests/host_cpp/bench_boundary.bin) of --config pass")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="time budget of the CPU baseline leg")
    ap.add_argument("--no-shard-rehearsal", action="store_true",
                    help="cfg3 / cfg5 at N = 1: skip the in-process 2 / 4 / 8-shard solves reported beside `value`")
    args = ap.parse_args()
    # --gpus N > 1 without a launcher: start one rank per GPU ourselves (python -m torch.distributed.run, rendezvous on
    # 127.0.0.1) BEFORE this process touches a GPU, relay the ranks' output and exit with their status.  On a box with
    # fewer GPUs than ranks the landmark shards run in ONE process on one GPU instead (rs_comm_init_local).
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        ndev = visible_gpu_count()              # from sysfs / the environment: no HIP call in this process
        if ndev >= args.gpus:
            import socket
            import subprocess
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            sys.exit(subprocess.run(cmd, env=env).returncode)
        args.in_process_shards = args.gpus
        if args.config in ("cfg2", "cfg4"):
            raise SystemExit("--gpus %d on a box with %d GPU(s): %s shards with no exchange step, there is nothing to rehearse "
                             "in one process; run it with --gpus 1" % (args.gpus, ndev, args.config))
        if args.config == "pass":
            args.config = "cfg3"                 # the pass's only exchange step is the sharded local BA
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
