"""GPU tests added in round 4: the folded set-up launches of the bundle adjustment (their "known zero" account of the
workspace), the lanes mode's stream ordering, the sharded solve's fast paths.
PARITY UNPINNED against the reference itself (it ships no fixtures): the oracle is the restatement in oracle/*.c.
"""
import numpy as np
import pytest

from conftest import to_np

pytestmark = pytest.mark.gpu


def _solve(ctx, w):
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
    return s, [t["outcome"] for t in ctx.ba_trace()], to_np(dc), to_np(dp)


def test_bundle_adjust_setup_launches_survive_workspace_reuse(ctx, rs, oracle, synth):
    """Round 4 folds K0 + the landmark grouping into two launches whose count kernel ADDS into a histogram that the previous
    solve's finalize kernel left at zero (csrc/ba.hip, `known_zero`).  The account must hold when the layout changes between
    solves, when another entry point scribbles over the workspace (tracks, a batch of windows) and on a fresh context."""
    wa = synth.make_ba_window(n_kf=12, n_points=3000, run_max=8, config_id=31)
    wb = synth.make_ba_window(n_kf=9, n_points=2600, run_max=6, config_id=32)
    rc, rp, os_, otr = None, None, None, None
    rca, rpa, osa = oracle.bundle_adjust(wa["cams"], wa["cam_free"], wa["points"], wa["obs_ptr"], wa["obs_cam"], wa["obs_uv"], wa["K"])
    first = _solve(ctx, wa)
    assert (first[0]["iterations"], first[0]["successful_steps"]) == (osa["iterations"], osa["successful_steps"])
    assert np.allclose(first[2], rca, rtol=1e-6, atol=1e-8)
    seq = [_solve(ctx, wa), _solve(ctx, wb), _solve(ctx, wa)]
    # another user of the workspace between two solves: the track stage and a batch of windows
    tk = synth.make_tracks(n_tracks=1500)
    ctx.triangulate_tracks(ctx.dev(tk["track_uv"]), ctx.dev(tk["sight_ptr"]), ctx.dev(tk["sight_pose"]), ctx.dev(tk["sight_uv"]),
                           ctx.dev(tk["poses"]), tk["kf_pose"], tk["K"], d_skip=ctx.dev(tk["skip"]))
    seq.append(_solve(ctx, wa))
    small = [synth.make_ba_window(n_kf=6, n_points=300 + 40 * i, run_max=5, config_id=90 + i) for i in range(3)]
    probs = [(ctx.dev(w["cams"]), w["cam_free"], ctx.dev(w["points"]), ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
             for w in small]
    ctx.bundle_adjust_batch(probs)
    seq.append(_solve(ctx, wa))
    for i in (0, 2, 3, 4):
        s, tr, c, p = seq[i]
        assert tr == first[1] and s["iterations"] == first[0]["iterations"], i
        assert np.isclose(s["final_cost"], first[0]["final_cost"], rtol=1e-10), i
        assert np.allclose(c, first[2], rtol=1e-8, atol=1e-10) and np.allclose(p, first[3], rtol=1e-7, atol=1e-9), i
    rcb, rpb, osb = oracle.bundle_adjust(wb["cams"], wb["cam_free"], wb["points"], wb["obs_ptr"], wb["obs_cam"], wb["obs_uv"], wb["K"])
    assert np.allclose(seq[1][2], rcb, rtol=1e-6, atol=1e-8) and seq[1][0]["iterations"] == osb["iterations"]
    c2 = rs.Context(0)
    try:
        s, tr, c, p = _solve(c2, wa)
        assert tr == first[1] and np.allclose(c, first[2], rtol=1e-8, atol=1e-10)
    finally:
        c2.close()


def test_bundle_adjust_batch_lanes_are_ordered_on_the_parent_stream(ctx, synth):
    """in lanes mode every window runs on a child context's stream, and ba_finalize raises its host
    flag before its device-side copies into d_cameras / d_points have finished.  rs_bundle_adjust_batch now joins the lane
    streams into the parent's before it returns: a device-side read on the parent stream straight after the call (no
    synchronisation in between) sees the finished results."""
    import torch
    ws = [synth.make_ba_window(n_kf=20, n_points=9000 + 300 * i, config_id=3 + i) for i in range(3)]      # 0.2 MB of points each
    single = []
    for w in ws:
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
        single.append((to_np(dc), to_np(dp)))
    ctx.set_int("ba_batch_mode", 1)
    try:
        for rep in range(3):
            probs, keep = [], []
            for w in ws:
                bc, bp = ctx.dev(w["cams"]), ctx.dev(w["points"])
                probs.append((bc, w["cam_free"], bp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"]))
                keep.append((bc, bp))
            torch.cuda.synchronize()
            ctx.bundle_adjust_batch(probs)
            copies = [(bc.clone(), bp.clone()) for bc, bp in keep]      # enqueued on the parent stream, nothing waited for
            for (cc, cp), (c1, p1) in zip(copies, single):
                assert np.allclose(to_np(cc), c1, rtol=1e-8, atol=1e-10) and np.allclose(to_np(cp), p1, rtol=1e-7, atol=1e-9), rep
    finally:
        ctx.set_int("ba_batch_mode", 0)


def _sharded_solves(rs, synth, w, bounds, ints):
    """The window solved unsharded on a context of its own, then as len(bounds) - 1 landmark shards on an in-process group
    (rs_comm_init_local), every context with the settings `ints`; returns (single, [per rank], shards)."""
    import threading
    import torch
    n_shards = len(bounds) - 1
    single = rs.Context(0)
    for k, v in ints.items():
        single.set_int(k, v)
    dc, dp = single.dev(w["cams"]), single.dev(w["points"])
    single.prof_begin()
    s0 = single.bundle_adjust(dc, w["cam_free"], dp, single.dev(w["obs_ptr"]), single.dev(w["obs_cam"]), single.dev(w["obs_uv"]), w["K"])
    ref = (s0, single.ba_trace(), to_np(dc), to_np(dp), single.prof_end())
    single.close()
    ctxs = [rs.Context(0) for _ in range(n_shards)]
    streams = [torch.cuda.Stream(device=ctxs[0].device) for _ in range(n_shards)]
    for c, st in zip(ctxs, streams):
        c.use_stream(st)
        for k, v in ints.items():
            c.set_int(k, v)
    rs.Context.comm_init_local(ctxs)
    shards = [synth.shard_ba_by_landmark(w, n_shards, r, bounds=bounds) for r in range(n_shards)]
    out = [None] * n_shards

    def work(r):
        try:
            c, sh = ctxs[r], shards[r]
            with torch.cuda.stream(streams[r]):
                dcr, dpr = c.dev(sh["cams"]), c.dev(sh["points"])
                args = (c.dev(sh["obs_ptr"]), c.dev(sh["obs_cam"]), c.dev(sh["obs_uv"]))
                streams[r].synchronize()
                c.prof_begin()
                s = c.bundle_adjust(dcr, sh["cam_free"], dpr, *args, sh["K"])
                prof = c.prof_end()
                out[r] = (s, c.ba_trace(), to_np(dcr), to_np(dpr), prof, c.ba_stats())
        except Exception as ex:      # noqa: BLE001
            out[r] = ex

    threads = [threading.Thread(target=work, args=(r,)) for r in range(n_shards)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    stuck = any(t.is_alive() for t in threads)
    if not stuck:
        for c in ctxs:
            c.comm_destroy()
            c.close()
    assert not stuck, "a rank is stuck in an exchange step"
    for r in range(n_shards):
        assert not isinstance(out[r], Exception), out[r]
    return ref, out, shards


def _check_shards(ref, out, shards, tol):
    s0, tr0, c0, p0, _ = ref
    for r, (s, tr, cr, pr, _, _) in enumerate(out):
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (s0["iterations"], s0["successful_steps"], s0["termination"], s0["usable"]), r
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in tr0], r
        assert np.isclose(s["final_cost"], s0["final_cost"], rtol=tol), r
        assert np.allclose(cr, c0, rtol=tol, atol=tol * 1e-2), r
        lo, hi = shards[r]["point_range"]
        assert np.allclose(pr, p0[lo:hi], rtol=tol, atol=tol * 1e-1), r
        if r > 0:
            assert np.array_equal(cr, out[0][2]), "ranks must end with bit-identical cameras (redundant reduced solves)"


@pytest.mark.parametrize("bounds", [(0, 450, 900), (0, 300, 300, 900), (0, 100, 220, 330, 450, 560, 680, 790, 900)])
def test_sharded_bundle_adjust_keeps_the_banded_reduced_solve(rs, synth, bounds):
    """a landmark-sharded solve of a block-banded window (n = 186 > 126: the blocked solver's domain) must run
    the same ONE-launch banded factorisation as the unsharded solve — the span of the whole window is agreed on by a MIN
    all-reduce of one key, after which every rank factors the same all-reduced system.  2, 3 (one EMPTY) and 8 shards on the
    in-process group against the unsharded solve: same kernels, same schedule, cameras bit-identical across ranks.
    Reference: one solver whatever the window (src/Optimization.cpp:127-142,360)."""
    w = synth.make_ba_window(n_kf=33, n_points=900, run_min=2, run_max=10, config_id=142)
    ref, out, shards = _sharded_solves(rs, synth, w, bounds, {})
    assert "K7b_band_factor" in ref[4]
    for r in range(len(out)):
        assert "K7b_band_factor" in out[r][4], (r, sorted(out[r][4]))
    _check_shards(ref, out, shards, 1e-7)


@pytest.mark.parametrize("bounds", [(0, 5000, 10000), (0, 3300, 3300, 10000), (0, 2000, 5500, 10000)])
def test_sharded_bundle_adjust_keeps_the_fused_solve_launch(rs, synth, bounds):
    """a landmark shard of a local window runs K7 + K8 as ONE launch between the round's two exchange steps
    (forced here with "ba_fuse_mode" 2: several ranks share this box's one GPU, where the default would keep the launches
    apart), incl. an EMPTY shard, which has no K8 workgroups and keeps the two launches.  The benchmark window (6 of its 10
    steps rejected: kept U / gc, three radii per round) against the unsharded solve."""
    w = synth.make_ba_window()
    ref, out, shards = _sharded_solves(rs, synth, w, bounds, {"ba_fuse_mode": 2})
    assert "K78_ba_solve_backsub" in ref[4]
    for r in range(len(out)):
        lo, hi = shards[r]["point_range"]
        assert ("K78_ba_solve_backsub" in out[r][4]) == (hi > lo), (r, sorted(out[r][4]))
        assert out[r][5]["handoff_retries"] == out[0][5]["handoff_retries"]      # a lost hand-off is re-run by EVERY rank or by none
    _check_shards(ref, out, shards, 1e-9)


@pytest.mark.parametrize("kw", [dict(n_kf=10, n_points=1500, run_max=7, config_id=201, outlier_frac=0.15, rot_noise_deg=3.0),
                                dict(n_kf=14, n_points=2500, run_max=9, config_id=202, outlier_frac=0.02, rot_noise_deg=0.2, pixel_noise=0.3),
                                dict(n_kf=7, n_points=600, run_max=5, config_id=203, outlier_frac=0.25, rot_noise_deg=4.0),
                                dict(n_kf=20, n_points=6000, config_id=204, outlier_frac=0.08, rot_noise_deg=1.5),
                                dict(n_kf=9, n_points=1200, run_max=6, config_id=205, outlier_frac=0.0, rot_noise_deg=0.1, pixel_noise=0.2),
                                dict(n_kf=16, n_points=4000, run_max=10, config_id=206, outlier_frac=0.12, rot_noise_deg=2.5)])
def test_speculation_depth_policy_keeps_the_schedule(ctx, oracle, synth, kw):
    """Round 4's depth policy (one radius where a step is most likely accepted, five while the trust region is uncalibrated,
    three afterwards, never more than the iterations left; csrc/ba_common.h, ba_decide) on windows with different accept /
    reject patterns — long rejection streaks, clean convergence, early termination: the per-iteration record must be the
    oracle's, and the rounds must be fewer than the iterations wherever a step was rejected."""
    w = synth.make_ba_window(**kw)
    _, _, os_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    rc, rp, _ = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
    tr, st = ctx.ba_trace(), ctx.ba_stats()
    assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
           (os_["iterations"], os_["successful_steps"], os_["termination"], os_["usable"])
    assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
    assert np.allclose([t["radius"] for t in tr], [t["radius"] for t in otr], rtol=1e-7)
    assert np.allclose([t["cost"] for t in tr], [t["cost"] for t in otr], rtol=1e-8)
    assert np.allclose(to_np(dc), rc, rtol=1e-6, atol=1e-8)
    out = [t["outcome"] for t in otr]
    # a rejected step FOLLOWED by another iteration shares its round with it (a rejection in the very last iteration has a
    # round of its own: the cap "never more radii than iterations left")
    shared = any(o in (0, -1) for o in out[:-1])
    assert st["rounds"] <= s["iterations"] and (not shared or st["rounds"] < s["iterations"]), (st, out)
    assert st["set_evaluations"] >= s["iterations"] - (1 if s["termination"] != 0 else 0)


@pytest.mark.parametrize("n_points,config_id,resident", [(11400, 3, 2), (17000, 35, 1), (17000, 36, 1)])
def test_fused_solve_launch_with_fewer_resident_radii(ctx, oracle, synth, n_points, config_id, resident):
    """The fused K7 + K8 launch holds the K8 workgroups of 3 speculative radii up to 10.7 k landmarks, of 2 up to 16 k, of 1 up to
    32 k (csrc/ba_solve.hip, ba_backsub_resident_sets): a K8 workgroup then evaluates radius s + rs, s + 2 rs, ... after radius s.
    Windows whose schedules have rejection streaks (so that the later passes' results are the ones taken): the launch is the
    fused one, the schedule is the oracle's step for step, the result the oracle's."""
    w = synth.make_ba_window(n_points=n_points, config_id=config_id)
    _, _, os_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    rc, rp, _ = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    assert [t["outcome"] for t in otr].count(0) >= 3                  # (the window was chosen for its rejected steps)
    ctx.set_int("ba_fuse_mode", 2)
    try:
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        ctx.prof_begin()
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
        prof = ctx.prof_end()
        tr = ctx.ba_trace()
    finally:
        ctx.set_int("ba_fuse_mode", 0)
    assert "K78_ba_solve_backsub" in prof and "K8_ba_backsub_cost" not in prof
    per = 128                                                           # landmarks per K8 workgroup of the fused launch
    assert 5 + (resident + 1) * ((n_points + per - 1) // per) > 256 >= 5 + resident * ((n_points + per - 1) // per)
    assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
    assert (s["iterations"], s["successful_steps"], s["termination"]) == (os_["iterations"], os_["successful_steps"], os_["termination"])
    assert np.allclose([t["cost"] for t in tr], [t["cost"] for t in otr], rtol=1e-9)
    assert np.isclose(s["final_cost"], os_["final_cost"], rtol=1e-10)
    assert np.allclose(to_np(dc), rc, rtol=1e-7, atol=1e-9)
    # (a few landmarks of such a window are seen along nearly parallel rays: their V has condition 1e10 and they move by 1e-4
    # for camera differences of 1e-14; tools/fuse_soak.py counts them)
    assert np.mean(np.abs(to_np(dp) - rp).max(axis=1) < 1e-6) > 0.995
