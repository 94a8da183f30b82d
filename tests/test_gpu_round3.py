"""GPU parity tests added in round 3 (VERDICT r2 "next round" items 1, 2, 4, 7): the benchmark pass's OWN inputs against
the oracle, cfg 5 at full size value by value, one order rule for match_for_fuse on both product paths, the lost
hand-off retry of the fused solve launch, and the envelope holes of the multi-rank / many-camera BA.
PARITY UNPINNED against the reference itself (it ships no fixtures): the oracle is the restatement in oracle/*.c.
"""
import threading

import numpy as np
import pytest

from conftest import to_np
from test_gpu_parity import _reproj_case

pytestmark = pytest.mark.gpu


class _Env:
    pass


def _bench_env(ctx, rs, synth):
    import torch
    e = _Env()
    e.ctx, e.rs, e.synth, e.torch = ctx, rs, synth, torch
    e.world, e.rank, e.local_rank = 1, 0, 0
    return e


def test_bench_pass_inputs_against_oracle(ctx, rs, oracle, synth):
    """the reprojection matcher was never compared with the oracle at the size the metric runs it
    (2000 keypoints x 10 000 points; 621 / 8 884 eligible).  This builds bench.py's pass inputs (build_pass) and checks
    both K2 calls — eight-lane kernel, packed tree, one-lane kernel — the cfg-2 match + triangulation, the track stage and
    the whole pass's outputs against the oracle on exactly those inputs.  Reference: src/MapMatcher.cpp:45-98,165-175."""
    import bench
    one_pass, cpu_pass, meta = bench.build_pass(_bench_env(ctx, rs, synth))
    assert len(meta["frame_a"]["keypoints"]) == 2000 and len(meta["mp_a"]["positions"]) == 10000
    n_a = _reproj_case(ctx, oracle, rs, meta["frame_a"], meta["mp_a"], 0)          # match_key_frame
    n_b = _reproj_case(ctx, oracle, rs, meta["frame_b"], meta["mp_b"], 0)          # match_map (first call's matches taken out)
    assert n_a > 100 and n_b > 100
    assert int(meta["mp_a"]["eligible"].sum()) == meta["match_key_frame_points"] < meta["match_map_points"]
    for _ in range(3):
        one_pass()
    ok, report = bench.check_pass_parity(meta, cpu_pass, oracle)
    assert ok, report
    for k in ("match_descriptors.query", "match_key_frame.points", "match_map.points", "triangulate.xyz", "tracks.status",
              "tracks.required_cos", "tracks.accepted", "build_local_window.frames", "bundle_adjust.schedule"):
        assert report[k] is True, (k, report)
    # the side stages the bench line reports beside the pass
    meta["cull_stage"]()
    meta["refine_stage"]()
    got = meta["gpu_results"]()
    ref_c = meta["cpu_cull"](oracle)
    n = len(meta["cull_in"]["positions"])
    assert np.array_equal(got["cull"]["mean_err"][:n].view(np.uint32), ref_c["mean_err"].view(np.uint32))
    assert np.array_equal(got["cull"]["cull"][:n], ref_c["cull"])
    ref_cam, ref_s = meta["cpu_refine"](oracle)
    cam, s = got["refine"]
    assert (s["iterations"], s["successful_steps"], s["termination"]) == (ref_s["iterations"], ref_s["successful_steps"], ref_s["termination"])
    assert np.allclose(cam, ref_cam, rtol=1e-7, atol=1e-9)


def test_bundle_adjust_cfg5_full_size_against_oracle(ctx, oracle, synth):
    """BASELINE.json configs[4] at FULL size — 100 key frames, 80 k landmarks, ~480 k observations, reduced system n = 588
    (window-size-independent MFMA Schur kernel + the blocked reduced solve) — value by value against the oracle, with the
    per-iteration trace (the oracle solves it in ~8 s).  Reference: src/Optimization.cpp:269-374.
    Tolerances: identical schedule; costs 1e-9 / 1e-7; poses and points 1e-6 relative (f64 summation order; the reference
    rounds its results to f32 on write-back)."""
    w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5)
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    args = (ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]))
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
    tr = ctx.ba_trace()
    rc, rp, rs_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    assert s["usable"] == rs_["usable"] == 1
    assert (s["iterations"], s["successful_steps"], s["termination"]) == (rs_["iterations"], rs_["successful_steps"], rs_["termination"])
    assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
    # (x_norm of the late iterations: the two f64 trajectories are ~1e-8 apart after six accepted steps on a 16 k-norm state)
    for k, tol in (("radius", 1e-7), ("cost", 1e-9), ("candidate_cost", 1e-7), ("model_cost_change", 1e-6), ("x_norm", 1e-7)):
        assert np.allclose([t[k] for t in tr], [t[k] for t in otr], rtol=tol), k
    assert np.isclose(s["initial_cost"], rs_["initial_cost"], rtol=1e-12) and np.isclose(s["final_cost"], rs_["final_cost"], rtol=1e-7)
    c1, p1 = to_np(dc), to_np(dp)
    assert np.allclose(c1, rc, rtol=1e-6, atol=1e-8)
    assert np.allclose(p1, rp, rtol=1e-6, atol=1e-7)
    fixed = ~np.asarray(w["cam_free"]).astype(bool)
    assert fixed.sum() == 2 and np.array_equal(c1[fixed], w["cams"][fixed])
    assert s["final_cost"] < 0.15 * s["initial_cost"]
    # size-independent property kept from round 2: a second solve started from the result finds (almost) nothing left
    s2 = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
    assert s2["initial_cost"] == pytest.approx(s["final_cost"], rel=1e-9)
    assert s2["final_cost"] <= s2["initial_cost"] and s2["final_cost"] > 0.5 * s2["initial_cost"]


def test_match_for_fuse_shuffled_list_one_order_rule(ctx, rs, oracle, synth):
    """match_for_fuse (src/MapMatcher.cpp:117-127) loops over its vector, so points that tie for a
    keypoint are decided by LIST order.  The flattened path (the shim flattens the vector in order) and the resident path
    (rs_map_match with only_points) must both follow that one rule: same result for the same shuffled list, equal to the
    oracle on the list-ordered arrays.  Duplicated descriptors force distance ties between listed points."""
    from test_resident_map import Scene
    sc = Scene(ctx, rs, synth, n_kf=6, n_points=600, seed=9, dup_pairs=True)
    rng = np.random.default_rng(5)
    # crowd the points: many of them share positions' projections and identical descriptors -> ties for one keypoint
    for p in range(0, 600, 2):
        sc.set_position(p + 1, sc.pos[p] + np.float32(1e-4))
    kpm = (rng.random(len(sc.frame["keypoints"])) < 0.3).astype(np.uint8)
    matched = rng.choice(600, 40, replace=False)
    for trial in range(3):
        only = rng.permutation(600)[:300].astype(np.int32)                       # shuffled: NOT map order
        fr = dict(sc.frame, kp_matched=kpm)
        full = sc.flat(matched_points=matched)
        # the listed points as their own arrays IN LIST ORDER (what the shim builds from the caller's vector)
        optr = [0]
        okf, odesc = [], []
        for p in only:
            a, b = full["obs_ptr"][p], full["obs_ptr"][p + 1]
            okf += list(full["obs_kf"][a:b]); odesc += list(full["obs_desc"][a:b])
            optr.append(len(okf))
        sub = dict(full, positions=full["positions"][only], eligible=full["eligible"][only], obs_ptr=np.array(optr, np.int32),
                   obs_kf=np.array(okf, np.int32), obs_desc=np.array(odesc, np.int32))
        ref = oracle.reproj_match(fr, sub, replace=1)
        fv, k1 = ctx.make_frame_view(fr)
        mv, k2 = ctx.make_map_view(sub)
        flat = ctx.reproj_match(fv, mv, replace=1)
        n = int(to_np(flat["count"])[0])
        assert n == len(ref["match_kp"]) > 20
        assert np.array_equal(to_np(flat["match_kp"])[:n], ref["match_kp"]) and np.array_equal(to_np(flat["match_point"])[:n], ref["match_point"])
        mk, mpt = sc.map.match(sc.rframe, fr["pose"], sc.K, fr["width"], fr["height"], kp_matched=kpm, matched_points=matched,
                               only_points=only, replace=1)
        assert np.array_equal(mk, ref["match_kp"])
        assert np.array_equal(mpt, only[ref["match_point"]])                      # map slots of the list-order winners
        # the rule matters on this scene: map order would have decided at least one keypoint differently
        if trial == 0:
            srt = np.sort(only)
            mk2, mpt2 = sc.map.match(sc.rframe, fr["pose"], sc.K, fr["width"], fr["height"], kp_matched=kpm, matched_points=matched,
                                     only_points=srt, replace=1)
            assert np.array_equal(mk2, mk)
            assert not np.array_equal(mpt2, mpt), "scene has no ties between listed points: the test would not see the order rule"
    # an empty list, and a list holding a dead slot (the reference skips null pointers)
    mk, mpt = sc.map.match(sc.rframe, sc.frame["pose"], sc.K, sc.frame["width"], sc.frame["height"], only_points=np.zeros(0, np.int32), replace=1)
    assert len(mk) == 0
    sc.remove_point(int(only[0]))
    mk, mpt = sc.map.match(sc.rframe, sc.frame["pose"], sc.K, sc.frame["width"], sc.frame["height"], only_points=only, replace=1)
    assert int(only[0]) not in set(mpt.tolist())
    sc.check(oracle)                                                              # the flag table was left clean
    sc.map.close()


def test_bundle_adjust_lost_handoff_is_rerun_as_two_launches(ctx, oracle, synth):
    """ADVICE r2 (medium) / VERDICT r2 #7a: in the fused K7 + K8 launch a K8 workgroup that does not see its hand-off word
    in time used to turn the solve into RS_BA_FAILURE.  That is a scheduling event, not a solver failure: the solve is now
    re-run once as separate launches from its untouched inputs.  The hook: "ba_handoff_timeout_us" = 1 makes every K8
    workgroup give up (K7 needs ~30 us).  The result must be the two-launch result and the oracle's schedule, and
    rs_ba_get_stats must count the retry."""
    w = synth.make_ba_window()
    _, _, os_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    args = (ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]))

    def run():
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
        return s, ctx.ba_trace(), to_np(dc), to_np(dp), ctx.ba_stats()

    try:
        ctx.set_int("ba_fuse_mode", 1)
        s1, t1, c1, p1, st1 = run()                       # two launches: the reference result of this test
        ctx.set_int("ba_fuse_mode", 2)
        s2, t2, c2, p2, st2 = run()                       # fused, default timeout: no retry
        assert st2["handoff_retries"] == st1["handoff_retries"]
        ctx.set_int("ba_handoff_timeout_us", 1)
        s3, t3, c3, p3, st3 = run()                       # fused, every consumer times out -> re-run in two-launch form
    finally:
        ctx.set_int("ba_handoff_timeout_us", 4000)
        ctx.set_int("ba_fuse_mode", 0)
    assert st3["handoff_retries"] == st2["handoff_retries"] + 1
    for s, tr in ((s1, t1), (s2, t2), (s3, t3)):
        assert s["usable"] == 1
        assert (s["iterations"], s["successful_steps"], s["termination"]) == (os_["iterations"], os_["successful_steps"], os_["termination"])
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
    assert np.isclose(s3["final_cost"], s1["final_cost"], rtol=1e-10)
    assert np.allclose(c3, c1, rtol=1e-9, atol=1e-11) and np.allclose(p3, p1, rtol=1e-9, atol=1e-10)
    # and the default path afterwards is healthy
    s4, t4, c4, p4, st4 = run()
    assert s4["usable"] == 1 and st4["handoff_retries"] == st3["handoff_retries"]


@pytest.mark.parametrize("kw", [dict(), dict(n_kf=8, n_points=150, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0),
                                dict(n_kf=12, n_points=3000, run_max=8, config_id=19, outlier_frac=0.06),
                                dict(n_kf=20, n_points=10120, config_id=3)])
def test_bundle_adjust_round_in_one_launch(ctx, oracle, synth, kw):
    """ba_fuse_mode 3: K5, K7 and K8 of a round as ONE launch (csrc/ba_round.hip) — the item workgroups count themselves for
    K7 when their atomics are performed, K7 reads the accumulators with L1-bypassing loads, the same workgroups then
    back-substitute their own landmarks behind K7's hand-off words.  Same per-iteration record as the oracle and as the
    separate launches, results equal to summation-order level, and twelve runs all reproduce it (a stale read behind any
    of the hand-offs would not).  The last case is the largest window that is resident at once (253 workgroups)."""
    w = synth.make_ba_window(**kw)
    _, _, os_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    runs = []
    retries_before = ctx.ba_stats()["handoff_retries"]
    try:
        ctx.set_int("ba_speculative_sets", 3)       # (the one-launch round is built for up to three radii per round)
        for mode in (1,) + (3,) * 12:
            ctx.set_int("ba_fuse_mode", mode)
            dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
            ctx.prof_begin()
            s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
            prof = ctx.prof_end()
            runs.append((s, ctx.ba_trace(), to_np(dc), to_np(dp), prof))
    finally:
        ctx.set_int("ba_fuse_mode", 0)
        ctx.set_int("ba_speculative_sets", 0)
    assert "K578_ba_round" in runs[1][4] and "K5_ba_schur_mfma" not in runs[1][4]
    assert "K578_ba_round" not in runs[0][4]
    s1, t1, c1, p1, _ = runs[0]
    for s, tr, c, p, _ in runs:
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (os_["iterations"], os_["successful_steps"], os_["termination"], os_["usable"])
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
        assert np.allclose([t["cost"] for t in tr], [t["cost"] for t in otr], rtol=1e-9)
        assert np.isclose(s["final_cost"], s1["final_cost"], rtol=1e-10)
        # (points: the 10120-landmark window holds two-observation outlier landmarks that end up 3e4 units away — nearly
        # unconstrained along the ray, they amplify the 1e-12 summation-order noise to 1e-9 relative)
        assert np.allclose(c, c1, rtol=1e-9, atol=1e-11) and np.allclose(p, p1, rtol=1e-8, atol=1e-9)
    assert ctx.ba_stats()["handoff_retries"] == retries_before


def _sharded_solve(rs, synth, w, shards, inertial=None, knobs=()):
    """n contexts on device 0 joined by rs_comm_init_local, one landmark shard each (see test_gpu_parity)."""
    import torch
    n = len(shards)
    ctxs = [rs.Context(0) for _ in range(n)]
    for c in ctxs:
        for name, value in knobs:
            c.set_int(name, value)
    streams = [torch.cuda.Stream(device=ctxs[0].device) for _ in range(n)]
    for c, st in zip(ctxs, streams):
        c.use_stream(st)
    rs.Context.comm_init_local(ctxs)
    out = [None] * n

    def work(r):
        try:
            c, sh = ctxs[r], shards[r]
            with torch.cuda.stream(streams[r]):
                dcr, dpr = c.dev(sh["cams"]), c.dev(sh["points"])
                args = (c.dev(sh["obs_ptr"]), c.dev(sh["obs_cam"]), c.dev(sh["obs_uv"]))
                streams[r].synchronize()
                if inertial is None:
                    s = c.bundle_adjust(dcr, sh["cam_free"], dpr, *args, sh["K"])
                    out[r] = (s, c.ba_trace(), to_np(dcr), to_np(dpr))
                else:
                    s, v, b = c.bundle_adjust_inertial(dcr, sh["cam_free"], dpr, *args, sh["K"], inertial)
                    out[r] = (s, c.ba_trace(), to_np(dcr), to_np(dpr), v, b)
        except Exception as ex:      # noqa: BLE001
            out[r] = ex

    threads = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in the exchange step"
    for c in ctxs:
        c.comm_destroy()
        c.close()
    for o in out:
        assert not isinstance(o, Exception), o
    return out


def test_bundle_adjust_empty_landmark_shard(rs, synth):
    """VERDICT r2 missing #4: an empty shard in a multi-rank solve was refused.  Eight ranks on a small window meet it.
    Three shards, the middle one EMPTY: it must take part in every exchange step with zero contributions and end with
    the same cameras as the others and as the unsharded solve."""
    w = synth.make_ba_window(n_kf=8, n_points=900, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0)
    single = rs.Context(0)
    dc, dp = single.dev(w["cams"]), single.dev(w["points"])
    s0 = single.bundle_adjust(dc, w["cam_free"], dp, single.dev(w["obs_ptr"]), single.dev(w["obs_cam"]), single.dev(w["obs_uv"]), w["K"])
    tr0, c0, p0 = single.ba_trace(), to_np(dc), to_np(dp)
    single.close()
    bounds = [0, 500, 500, 900]
    shards = [synth.shard_ba_by_landmark(w, 3, r, bounds=bounds) for r in range(3)]
    assert len(shards[1]["points"]) == 0 and len(shards[1]["obs_cam"]) == 0
    out = _sharded_solve(rs, synth, w, shards)
    for r in range(3):
        s, tr, cr, pr = out[r]
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (s0["iterations"], s0["successful_steps"], s0["termination"], s0["usable"])
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in tr0]
        assert np.isclose(s["final_cost"], s0["final_cost"], rtol=1e-6)
        assert np.allclose(cr, c0, rtol=1e-6, atol=1e-8)
        lo, hi = shards[r]["point_range"]
        assert np.allclose(pr, p0[lo:hi], rtol=1e-6, atol=1e-7)
        assert np.array_equal(cr, out[0][2]), "ranks must end with bit-identical cameras"


@pytest.mark.parametrize("imu_mode", [0, 1])
def test_bundle_adjust_inertial_landmark_sharded(rs, oracle, synth, imu_mode):
    """VERDICT r2 missing #4: inertial factors in a landmark-sharded solve were refused.  The IMU blocks are camera-side:
    every rank adds them to the (already all-reduced) camera system redundantly, like it solves redundantly.  Two shards
    against the oracle's inertial solve: schedule, poses, velocities, biases."""
    w = synth.make_ba_window(n_kf=8, n_points=400, run_max=6, config_id=61)
    imu = synth.make_imu(w)
    rc, rp, rv, rb, rs_, otr = oracle.bundle_adjust_inertial(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"],
                                                             w["obs_uv"], w["K"], imu, trace=True)
    shards = [synth.shard_ba_by_landmark(w, 2, r) for r in range(2)]
    out = _sharded_solve(rs, synth, w, shards, inertial=imu, knobs=(("ba_imu_mode", imu_mode),))
    for r in range(2):
        s, tr, cr, pr, v, b = out[r]
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (rs_["iterations"], rs_["successful_steps"], rs_["termination"], rs_["usable"])
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
        assert np.isclose(s["final_cost"], rs_["final_cost"], rtol=1e-7)
        assert np.allclose(cr, rc, rtol=1e-6, atol=1e-8)
        lo, hi = shards[r]["point_range"]
        assert np.allclose(pr, rp[lo:hi], rtol=1e-6, atol=1e-7)
        assert np.allclose(v, rv, rtol=1e-6, atol=1e-8) and np.allclose(b, rb, rtol=1e-5, atol=1e-8)


def test_bundle_adjust_two_hundred_free_cameras(ctx, rs, oracle, synth):
    """VERDICT r2 missing #4: more than 182 free cameras were refused (the reference has no such limit,
    src/Optimization.cpp:269-374).  240 key frames, 238 free: the generic K5 with its camera partial sums in 80 KB of LDS,
    the blocked reduced solve at n = 1428; against the oracle."""
    w = synth.make_ba_window(n_kf=240, n_points=2500, run_min=2, run_max=8, config_id=55)
    assert int(np.sum(w["cam_free"])) == 238
    o = oracle.default_options(); o.max_num_iterations = 4
    go = rs.default_options(); go.max_num_iterations = 4
    rc, rp, rs_ = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"], o)
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"], go)
    assert s["usable"] == rs_["usable"] == 1
    assert (s["iterations"], s["successful_steps"], s["termination"]) == (rs_["iterations"], rs_["successful_steps"], rs_["termination"])
    assert np.isclose(s["final_cost"], rs_["final_cost"], rtol=1e-7)
    assert np.allclose(to_np(dc), rc, rtol=1e-6, atol=1e-8)
    assert np.allclose(to_np(dp), rp, rtol=1e-6, atol=1e-7)


def test_comm_count_and_empty_launch(ctx, rs):
    """rs_comm_count reports what the exchange step runs over (ncclCommCount for RCCL); rs_prof_empty_launch the launch
    latency bench.py prints next to its numbers."""
    assert ctx.comm_count() == (1, 0)
    ctx.comm_init(rs.Context.comm_unique_id(), 1, 0)
    try:
        assert ctx.comm_count() == (1, 1)
    finally:
        ctx.comm_destroy()
    us = ctx.empty_launch_us(500)
    assert 0.5 < us < 100.0


@pytest.mark.parametrize("n", [1, 2, 7, 64, 255, 256, 257, 2000])
def test_triangulate_host_fast_path(ctx, oracle, synth, n):
    """VERDICT r2 #6a: triangulate_points as the reference's callers use it (host vectors in / out; ONE correspondence per
    call in Mapper::triangulate_tracks, src/Mapper.cpp:253).  rs_triangulate_host runs n <= 256 as a single one-workgroup
    launch with the result in pinned memory behind a completion flag, larger n through the staging pool; both must give
    the oracle's list bit for bit (indices, f32 positions), for both gate settings, and repeated calls must not see a
    previous call's flag or result."""
    pr = synth.make_pair(2)
    mq, mt = oracle.match_descriptors(pr["desc2"], pr["desc1"])
    reps = max(1, -(-n // len(mq)))
    uv1 = np.tile(pr["kp1"][mt], (reps, 1))[:n]
    uv2 = np.tile(pr["kp2"][mq], (reps, 1))[:n]
    uv2 = uv2 + (np.arange(n)[:, None] % 5 == 4) * np.float32(9.0)      # every fifth correspondence is wrong: gates reject some
    for cos, err in ((0.9999, 2.0), (1.0, 4.0)):
        ref = oracle.triangulate(uv1, uv2, pr["poses"], pr["K"], None, None, cos, err)
        for _ in range(3):
            idx, xyz = ctx.triangulate_host(uv1, uv2, pr["poses"][0], pr["poses"][1], pr["K"], cos, err)
            assert np.array_equal(idx, ref["out_index"])
            assert np.array_equal(xyz.view(np.uint32), ref["out_xyz"].view(np.uint32))
    assert 0 < len(ref["out_index"]) or n < 5
    idx, xyz = ctx.triangulate_host(uv1[:0], uv2[:0], pr["poses"][0], pr["poses"][1], pr["K"])
    assert len(idx) == 0


@pytest.mark.parametrize("kw,banded", [(dict(n_kf=100, n_points=1500, run_min=2, run_max=10, config_id=140), True),
                                       (dict(n_kf=60, n_points=2500, run_min=2, run_max=10, config_id=141), True),      # n = 348: 5 full blocks + 28 columns
                                       (dict(n_kf=33, n_points=900, run_min=2, run_max=10, config_id=142), True),       # n = 186
                                       (dict(n_kf=34, n_points=900, run_min=2, run_max=10, config_id=144), True),       # n = 192: whole blocks, no padding
                                       (dict(n_kf=35, n_points=900, run_min=2, run_max=10, config_id=145), True),       # n = 198: 58 padded columns (side 1 starts in its last 8-column step)
                                       (dict(n_kf=43, n_points=1200, run_min=2, run_max=10, config_id=146), True),      # n = 246: 10 padded columns
                                       (dict(n_kf=40, n_points=4000, run_min=3, run_max=24, config_id=143), False)])    # spans up to 23: not banded
def test_bundle_adjust_banded_reduced_solve(ctx, oracle, synth, kw, banded):
    """VERDICT r2 #5 (cfg 5's reduced solve was 13 + 12 dependent launches per LM step): when every landmark is seen by key
    frames at most 9 slots apart, S is block-banded and is factorised by ONE launch (csrc/ba_solve_big.hip, ba_band_factor;
    the sparsity Ceres' SPARSE_SCHUR exploits, src/Optimization.cpp:360) — two workgroups eliminating towards a separator
    block from both ends (default) or one workgroup ("ba_band_mode" 2).  Against the oracle and against the general
    blocked factorisation ("ba_band_mode" 1), incl. a matrix size that ends in a partial block and a window that is NOT
    banded (must take the general path by itself)."""
    w = synth.make_ba_window(**kw)
    _, _, os_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    rc, rp, _ = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    runs = {}
    try:
        for mode in (0, 2, 1):
            ctx.set_int("ba_band_mode", mode)
            dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
            ctx.prof_begin()
            s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
            prof = ctx.prof_end()
            runs[mode] = (s, ctx.ba_trace(), to_np(dc), to_np(dp), prof)
    finally:
        ctx.set_int("ba_band_mode", 0)
    assert ("K7b_band_factor" in runs[0][4]) == banded and "K7b_band_factor" not in runs[1][4]
    assert ("K7b_band_factor" in runs[2][4]) == banded
    # (round 4) the two-sided form factors its separator block in a launch of its own; the one-workgroup form has none
    assert ("K7c_band_separator" in runs[0][4]) == banded and "K7c_band_separator" not in runs[2][4]
    for mode in (0, 2, 1):
        s, tr, c, p, _ = runs[mode]
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (os_["iterations"], os_["successful_steps"], os_["termination"], os_["usable"]), mode
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr], mode
        assert np.allclose([t["cost"] for t in tr], [t["cost"] for t in otr], rtol=1e-7), mode      # (two f64 trajectories from a far-off start: 2e-9 by the fifth step)
        assert np.isclose(s["final_cost"], os_["final_cost"], rtol=1e-7), mode
        if banded:
            assert np.allclose(c, rc, rtol=1e-6, atol=1e-8) and np.allclose(p, rp, rtol=1e-6, atol=1e-7), mode
        else:
            # The long-track window is ILL-CONDITIONED, and its tolerance is an absolute one derived from
            # measurement (tools/tol_check.py, round 4): the Jacobi-scaled reduced camera system at the solution is singular
            # to working precision undamped (smallest eigenvalue -6e-8 against 2.2: the 40-frame chain of long tracks is
            # almost free along its gauge directions) and has condition 7.3e6 at the final radius (3.3e6).  Four GPU runs of
            # the SAME kernels differ by up to 3.3e-7 on a camera entry and 1.7e-5 on a point coordinate (the order of the
            # f64 atomics, amplified over ten LM steps); GPU against oracle: 4.4e-7 / 9.0e-6.  Asserted at ~10x those figures.
            assert np.abs(c - rc).max() < 5e-6 and np.abs(p - rp).max() < 1e-4, (mode, np.abs(c - rc).max(), np.abs(p - rp).max())
    if banded:
        assert np.allclose(runs[0][2], runs[1][2], rtol=1e-8, atol=1e-10)
        assert np.allclose(runs[0][2], runs[2][2], rtol=1e-8, atol=1e-10)      # two-sided against the one-workgroup form
    else:       # the same kernels both times: run-to-run noise of the ill-conditioned window (measured 1.1e-7 .. 3.3e-7)
        assert np.abs(runs[0][2] - runs[1][2]).max() < 5e-6


@pytest.mark.parametrize("bounds", [(0, 1000, 2000, 3000), (0, 1700, 1700, 3000), (0, 5, 2990, 3000)])
def test_reproj_match_sharded_map_min_reduction(rs, oracle, synth, bounds):
    """SURVEY.md 8(e) row 2 / VERDICT r2 #10: the map sharded over ranks, one min-reduction of the packed per-keypoint
    proposals (distance << 32 | global map order) before the accept step.  Three contexts of one process joined by
    rs_comm_init_local (on-device minimum instead of ncclAllReduce(min, u64)), uneven shards incl. an EMPTY one: every rank
    must return the unsharded result and the oracle's, bit for bit — ties between points of different shards included
    (duplicated descriptors force them).  Reference tie rule: src/MapMatcher.cpp:95-97."""
    import torch
    w = synth.make_ba_window(n_kf=10, n_points=3000)
    frame, mp = synth.make_match_scene(w, n_keypoints=1500, kdtree_build=rs.kdtree_build)
    rng = np.random.default_rng(17)
    # duplicated descriptors across the map: distance ties between points that live on different shards
    mp["desc_pool"][:] = mp["desc_pool"][rng.integers(0, 60, len(mp["desc_pool"]))]
    frame["descriptors"][:] = mp["desc_pool"][rng.integers(0, 60, len(frame["descriptors"]))]
    ref = oracle.reproj_match(frame, mp, replace=0)
    assert len(ref["match_kp"]) > 50
    P = len(mp["positions"])
    n = len(bounds) - 1
    ctxs = [rs.Context(0) for _ in range(n)]
    streams = [torch.cuda.Stream(device=ctxs[0].device) for _ in range(n)]
    for c, st in zip(ctxs, streams):
        c.use_stream(st)
    rs.Context.comm_init_local(ctxs)
    out = [None] * n

    def shard(lo, hi):
        o0, o1 = int(mp["obs_ptr"][lo]), int(mp["obs_ptr"][hi])
        return dict(mp, positions=mp["positions"][lo:hi], eligible=mp["eligible"][lo:hi], obs_ptr=(mp["obs_ptr"][lo:hi + 1] - o0).astype(np.int32),
                    obs_kf=mp["obs_kf"][o0:o1] if o1 > o0 else np.zeros(1, np.int32), obs_desc=mp["obs_desc"][o0:o1] if o1 > o0 else np.zeros(1, np.int32))

    def work(r):
        try:
            c = ctxs[r]
            with torch.cuda.stream(streams[r]):
                lo, hi = bounds[r], bounds[r + 1]
                sh = shard(lo, hi)
                if hi == lo:
                    sh["positions"] = np.zeros((0, 3), np.float32); sh["eligible"] = np.zeros(0, np.uint8); sh["obs_ptr"] = np.zeros(1, np.int32)
                fv, k1 = c.make_frame_view(frame, pack=True)
                mv, k2 = c.make_map_view(sh)
                mv.n_points = hi - lo
                streams[r].synchronize()
                o = c.reproj_match_sharded(fv, mv, lo)
                streams[r].synchronize()
                cnt = int(to_np(o["count"])[0])
                out[r] = (to_np(o["match_kp"])[:cnt].copy(), to_np(o["match_point"])[:cnt].copy(), to_np(o["prop_point"])[:1500].copy(),
                          to_np(o["point_kp"])[:hi - lo].copy())
        except Exception as ex:      # noqa: BLE001
            out[r] = ex

    threads = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in the exchange step"
    for c in ctxs:
        c.comm_destroy()
        c.close()
    for r in range(n):
        assert not isinstance(out[r], Exception), out[r]
        mk, mpt, prop, pkp = out[r]
        assert np.array_equal(mk, ref["match_kp"]) and np.array_equal(mpt, ref["match_point"]), r
        assert np.array_equal(prop, ref["prop_point"]), r
        assert np.array_equal(pkp, ref["point_kp"][bounds[r]:bounds[r + 1]]), r
    assert P == bounds[-1]


def test_context_fork_and_wait_for_order_streams(rs, oracle, synth):
    """rs_context_fork / rs_context_wait_for: stream ordering between contexts without the host.  Producer -> consumer across
    two contexts on streams of their own (the match list of rs_match_descriptors feeds rs_triangulate_matches on the other
    context), behind a long-running kernel sequence on the producer's stream so that an unordered consumer would read the
    list too early; then the reverse edge, many times.  Same stream / same context are no-ops; bad arguments are refused."""
    import torch
    a, b = rs.Context(0), rs.Context(0)
    sa, sb = torch.cuda.Stream(device=a.device), torch.cuda.Stream(device=a.device)
    a.use_stream(sa)
    b.use_stream(sb)
    try:
        pr = synth.make_pair(2)
        nq, nt = len(pr["desc2"]), len(pr["desc1"])
        dq, dt = a.dev(pr["desc2"]), a.dev(pr["desc1"])
        dk1, dk2, dpo = a.dev(pr["kp1"]), a.dev(pr["kp2"]), a.dev(pr["poses"])
        torch.cuda.synchronize()
        mq, mt = oracle.match_descriptors(pr["desc2"], pr["desc1"])
        ref = oracle.triangulate(pr["kp1"][mt], pr["kp2"][mq], pr["poses"], pr["K"])
        m = a.match_descriptors(dq, dt, nq, nt)
        t = b.triangulate_matches(dk1, dk2, m["mt"], m["mq"], m["cnt"], nq, dpo, pr["K"])
        torch.cuda.synchronize()
        for rep in range(10):
            for o in (m["mq"], m["mt"], m["cnt"], t["out_index"], t["count"]):
                o.zero_()
            torch.cuda.synchronize()
            b.fork(a)                                  # a's work waits for b's (nothing there yet): exercises the edge b -> a
            for _ in range(6):                         # ~60 us of matching in front of the list the consumer needs
                a.match_descriptors(dq, dt, nq, nt, out=m)
            if rep % 2:
                b.wait_for(a)
            else:
                a.fork(b)
            b.triangulate_matches(dk1, dk2, m["mt"], m["mq"], m["cnt"], nq, dpo, pr["K"], out=t)
            a.wait_for(b)
            a.wait_for(a, b)                           # own stream: skipped
            torch.cuda.synchronize()
            cnt = int(to_np(t["count"])[0])
            assert cnt == len(ref["out_index"]) and cnt > 100, rep
            assert np.array_equal(to_np(t["out_index"])[:cnt], ref["out_index"]), rep
            assert np.array_equal(to_np(t["out_xyz"])[:cnt].view(np.uint32), ref["out_xyz"].view(np.uint32)), rep
        assert a.lib.rs_context_wait_for(a.h, None, 2) != 0 and a.lib.rs_context_fork(None, None, 0) != 0
    finally:
        torch.cuda.synchronize()
        a.close()
        b.close()


def test_bundle_adjust_item_size_and_decision_launch(ctx, oracle, synth):
    """"ba_item_landmarks" (32 ... 64 landmarks per workgroup of K5, in steps of 8) and the round's decision as a launch of its own in front
    of K5 (taken when a solve has more items than compute units: 9000 landmarks in items of 32 = 282 items): the LM schedule
    and the results do not depend on either (atomic order only)."""
    w = synth.make_ba_window(n_kf=12, n_points=9000, config_id=151)
    _, _, os_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    rc, rp, _ = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    runs = {}
    try:
        for item in (0, 32, 40, 48, 56, 64):
            ctx.set_int("ba_item_landmarks", item)
            dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
            s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
            runs[item] = (s, ctx.ba_trace(), to_np(dc), to_np(dp))
    finally:
        ctx.set_int("ba_item_landmarks", 0)
    for item, (s, tr, c, p) in runs.items():
        assert (s["iterations"], s["successful_steps"], s["termination"]) == (os_["iterations"], os_["successful_steps"], os_["termination"]), item
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr], item
        assert np.allclose([t["cost"] for t in tr], [t["cost"] for t in otr], rtol=1e-8), item
        assert np.allclose(c, rc, rtol=1e-7, atol=1e-9) and np.allclose(p, rp, rtol=1e-7, atol=1e-8), item
    with pytest.raises(Exception):
        ctx.set_int("ba_item_landmarks", 44)
