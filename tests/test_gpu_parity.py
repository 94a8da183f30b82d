"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on
identical seeded inputs.  Integer / index results are compared bit for bit;
floating point results within the tolerance written next to each assert.
PARITY UNPINNED against the reference itself (it ships no fixtures): the oracle
is the restatement documented in oracle/*.c.
"""
import numpy as np
import pytest

from conftest import to_np

pytestmark = pytest.mark.gpu


def _knn_case(ctx, oracle, q, t):
    nq, nt = len(q), len(t)
    dq, dt = ctx.dev(q), ctx.dev(t)
    got = [to_np(x)[0, :nq] for x in ctx.hamming_knn2(dq, dt, nq, nt)]
    ref = oracle.hamming_knn2(q, t)
    for g, r, name in zip(got, ref, ("idx0", "dist0", "idx1", "dist1")):
        assert np.array_equal(g, r), name
    m = ctx.match_descriptors(dq, dt, nq, nt)
    cnt = int(to_np(m["cnt"])[0])
    rq, rt = oracle.match_descriptors(q, t)
    assert cnt == len(rq)
    assert np.array_equal(to_np(m["mq"])[0, :cnt], rq)
    assert np.array_equal(to_np(m["mt"])[0, :cnt], rt)
    return cnt


@pytest.mark.parametrize("config_id", [1, 2])
def test_match_descriptors_configs(ctx, oracle, synth, config_id):
    pr = synth.make_pair(config_id)
    cnt = _knn_case(ctx, oracle, pr["desc2"], pr["desc1"])
    assert cnt > 0.5 * len(pr["desc2"]) * 0.8


@pytest.mark.parametrize("nq,nt", [(1, 1), (7, 1), (1, 2), (65, 33), (130, 77), (64, 4096), (1000, 3)])
def test_knn2_ragged_sizes(ctx, oracle, nq, nt):
    rng = np.random.default_rng(nq * 1000 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    _knn_case(ctx, oracle, q, t)


def test_knn2_ties_and_thresholds(ctx, oracle):
    """equal distances (lower train index must win), d0 == 64 (accepted), 4*d0 == 3*d1 (accepted)."""
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    t = np.zeros((300, 32), np.uint8)
    for i in range(300):
        src = q[i % 200].copy()
        bits = np.unpackbits(src)
        flip = rng.choice(256, size=[0, 3, 48, 64, 64, 65, 85][i % 7], replace=False)
        bits[flip] ^= 1
        t[i] = np.packbits(bits)
    t[100:110] = t[0]           # exact duplicates: ties at every rank
    q[150:160] = q[0]
    _knn_case(ctx, oracle, q, t)


def test_match_descriptors_batched(ctx, oracle, synth):
    prs = [synth.make_pair(1, seed_stream=s) for s in range(3)]
    q = np.stack([p["desc2"] for p in prs])
    t = np.stack([p["desc1"] for p in prs])
    nq, nt = q.shape[1], t.shape[1]
    m = ctx.match_descriptors(ctx.dev(q), ctx.dev(t), nq, nt, batch=3, raw=True)
    for b in range(3):
        rq, rt = oracle.match_descriptors(q[b], t[b])
        cnt = int(to_np(m["cnt"])[b])
        assert cnt == len(rq)
        assert np.array_equal(to_np(m["mq"])[b, :cnt], rq)
        assert np.array_equal(to_np(m["mt"])[b, :cnt], rt)
        ref = oracle.hamming_knn2(q[b], t[b])
        for g, r in zip(m["raw"], ref):
            assert np.array_equal(to_np(g)[b], r)


def test_match_descriptors_empty(ctx):
    import torch
    q = torch.zeros((4, 32), dtype=torch.uint8, device=ctx.device)
    m = ctx.match_descriptors(q, q, 4, 0)
    assert int(to_np(m["cnt"])[0]) == 0
    m = ctx.match_descriptors(q, q, 0, 4)
    assert int(to_np(m["cnt"])[0]) == 0


# ------------------------------------------------------------ triangulation
def _tri_case(ctx, oracle, uv1, uv2, poses, K, idx1=None, idx2=None, cos=0.9999, err=2.0):
    n = len(uv1)
    d = ctx.triangulate(ctx.dev(uv1, np.float32), ctx.dev(uv2, np.float32), n, ctx.dev(poses, np.float32),
                        len(poses), K, None if idx1 is None else ctx.dev(idx1, np.int32),
                        None if idx2 is None else ctx.dev(idx2, np.int32), cos, err)
    ref = oracle.triangulate(uv1, uv2, poses, K, idx1, idx2, cos, err)
    xyz = to_np(d["xyz"])[:n]
    keep = to_np(d["keep"])[:n]
    # f32 tolerance: 1e-5 relative (the f64 SVD + f32 rounding is the same arithmetic on both
    # sides; with -ffp-contract=off the results are expected to be bit-identical)
    assert np.allclose(xyz, ref["xyz"], rtol=1e-5, atol=1e-6)
    exact = np.array_equal(xyz.view(np.uint32), ref["xyz"].view(np.uint32))
    assert np.array_equal(keep, ref["keep"]), "gate decisions differ"
    cnt = int(to_np(d["count"])[0])
    assert cnt == len(ref["out_index"])
    assert np.array_equal(to_np(d["out_index"])[:cnt], ref["out_index"])
    assert np.allclose(to_np(d["out_xyz"])[:cnt], ref["out_xyz"], rtol=1e-5, atol=1e-6)
    return exact, cnt


@pytest.mark.parametrize("config_id", [1, 2])
def test_triangulate_pair(ctx, oracle, synth, config_id):
    pr = synth.make_pair(config_id)
    mq, mt = oracle.match_descriptors(pr["desc2"], pr["desc1"])
    uv1, uv2 = pr["kp1"][mt], pr["kp2"][mq]
    exact, cnt = _tri_case(ctx, oracle, uv1, uv2, pr["poses"], pr["K"])
    assert cnt > 0
    assert exact, "positions are expected bit-exact with contraction off"
    # Mapper's gates (src/Mapper.cpp:37-38)
    _tri_case(ctx, oracle, uv1, uv2, pr["poses"], pr["K"], cos=1.0, err=4.0)


def test_triangulate_matches_device_list(ctx, oracle, synth):
    """rs_triangulate_matches: get_matching_points fused, match list + count stay on the device."""
    pr = synth.make_pair(1)
    nq = len(pr["desc2"])
    m = ctx.match_descriptors(ctx.dev(pr["desc2"]), ctx.dev(pr["desc1"]), nq, len(pr["desc1"]))
    d = ctx.triangulate_matches(ctx.dev(pr["kp1"]), ctx.dev(pr["kp2"]), m["mt"], m["mq"], m["cnt"], nq,
                                ctx.dev(pr["poses"]), pr["K"])
    mq, mt = oracle.match_descriptors(pr["desc2"], pr["desc1"])
    ref = oracle.triangulate(pr["kp1"][mt], pr["kp2"][mq], pr["poses"], pr["K"])
    n = len(mq)
    assert np.array_equal(to_np(d["keep"])[:n], ref["keep"])
    assert np.array_equal(to_np(d["xyz"])[:n].view(np.uint32), ref["xyz"].view(np.uint32))
    cnt = int(to_np(d["count"])[0])
    assert cnt == len(ref["out_index"]) and np.array_equal(to_np(d["out_index"])[:cnt], ref["out_index"])


def test_triangulate_tracks_per_item_poses(ctx, oracle, synth):
    w = synth.make_ba_window(n_kf=8, n_points=300, run_max=6)
    rng = np.random.default_rng(3)
    poses = w["poses_true"].reshape(-1, 16)
    first = w["obs_ptr"][:-1]
    last = w["obs_ptr"][1:] - 1
    idx1 = w["obs_cam"][first]
    idx2 = w["obs_cam"][last]
    uv1, uv2 = w["obs_uv"][first], w["obs_uv"][last]
    exact, cnt = _tri_case(ctx, oracle, uv1, uv2, poses, w["K"], idx1, idx2, cos=1.0, err=4.0)
    assert cnt > 100
    del rng


@pytest.mark.parametrize("kw,quota", [(dict(n_tracks=2000, config_id=6), 100),
                                      (dict(n_tracks=300, far_frac=0.9, config_id=7), 100),
                                      (dict(n_tracks=5000, n_frames=16, max_sightings=14, config_id=9), 100),
                                      (dict(n_tracks=37, config_id=10), 1000)])
def test_triangulate_tracks_body(ctx, oracle, synth, kw, quota):
    """K6 (body of Mapper::triangulate_tracks, src/Mapper.cpp:246-305) vs the oracle: statuses, points and parallax
    cosines bit for bit; `required` within device-libm ulps (acosf / cosf); the accepted / inconsistent lists
    identical unless a candidate sits within an ulp of its requirement."""
    sc = synth.make_tracks(**kw)
    ref = oracle.triangulate_tracks(sc["track_uv"], sc["sight_ptr"], sc["sight_pose"], sc["sight_uv"], sc["poses"],
                                    sc["kf_pose"], sc["K"], skip=sc["skip"], min_new_points=quota)
    d = ctx.triangulate_tracks(ctx.dev(sc["track_uv"]), ctx.dev(sc["sight_ptr"]), ctx.dev(sc["sight_pose"]),
                               ctx.dev(sc["sight_uv"]), ctx.dev(sc["poses"]), sc["kf_pose"], sc["K"],
                               d_skip=ctx.dev(sc["skip"]), min_new_points=quota)
    n = len(sc["track_uv"])
    assert np.array_equal(to_np(d["status"])[:n], ref["status"])
    assert np.array_equal(to_np(d["xyz"])[:n].view(np.uint32), ref["xyz"].view(np.uint32))
    assert np.array_equal(to_np(d["parallax_cos"])[:n].view(np.uint32), ref["parallax_cos"].view(np.uint32))
    req = to_np(d["required_cos"])[:n]
    assert np.allclose(req, ref["required_cos"], rtol=0, atol=3e-7)
    cnt = to_np(d["counts"])
    assert np.array_equal(to_np(d["inconsistent"])[:cnt[2]], ref["inconsistent"])
    cand = ref["status"] == 1
    boundary = cand & (np.abs(ref["parallax_cos"] - ref["required_cos"]) <= 3e-7)
    if not boundary.any():
        assert cnt[1] == ref["n_topped_up"]
        assert np.array_equal(to_np(d["accepted"])[:cnt[0]], ref["accepted"])
    else:       # decisions may differ only on the boundary tracks
        diff = set(to_np(d["accepted"])[:cnt[0]].tolist()) ^ set(ref["accepted"].tolist())
        assert len(diff) <= 2 * int(boundary.sum())
    # round 3: with the requirement table from the host's libm (rs_parallax_requirements — what the shims pass) `required`
    # and the accepted list are bit-identical, boundary tracks or not
    import importlib
    rsgpu = importlib.import_module("racing-slam_amd").rsgpu
    req = rsgpu.parallax_requirements(sc["poses"], sc["kf_pose"])
    e = ctx.triangulate_tracks(ctx.dev(sc["track_uv"]), ctx.dev(sc["sight_ptr"]), ctx.dev(sc["sight_pose"]),
                               ctx.dev(sc["sight_uv"]), ctx.dev(sc["poses"]), sc["kf_pose"], sc["K"],
                               d_skip=ctx.dev(sc["skip"]), min_new_points=quota, d_required=ctx.dev(req))
    ce = to_np(e["counts"])
    assert np.array_equal(to_np(e["status"])[:n], ref["status"])
    assert np.array_equal(to_np(e["required_cos"])[:n].view(np.uint32), ref["required_cos"].view(np.uint32))
    assert ce[1] == ref["n_topped_up"] and np.array_equal(to_np(e["accepted"])[:ce[0]], ref["accepted"])
    assert np.array_equal(to_np(e["inconsistent"])[:ce[2]], ref["inconsistent"])


def test_triangulate_tracks_body_edge_cases(ctx, synth):
    sc = synth.make_tracks(n_tracks=64, config_id=11)
    e = ctx.triangulate_tracks(ctx.dev(np.zeros((0, 2), np.float32)), ctx.dev(np.zeros(1, np.int32)),
                               ctx.dev(np.zeros(1, np.int32)), ctx.dev(np.zeros((1, 2), np.float32)), ctx.dev(sc["poses"]),
                               0, sc["K"])
    assert to_np(e["counts"]).tolist() == [0, 0, 0]
    z = ctx.triangulate_tracks(ctx.dev(sc["track_uv"]), ctx.dev(sc["sight_ptr"]), ctx.dev(sc["sight_pose"]),
                               ctx.dev(sc["sight_uv"]), ctx.dev(sc["poses"]), sc["kf_pose"], sc["K"],
                               d_skip=ctx.dev(np.ones(64, np.uint8)))
    assert to_np(z["counts"]).tolist() == [0, 0, 0] and not to_np(z["status"])[:64].any()
    with pytest.raises(Exception):
        ctx.triangulate_tracks(ctx.dev(sc["track_uv"]), ctx.dev(sc["sight_ptr"]), ctx.dev(sc["sight_pose"]),
                               ctx.dev(sc["sight_uv"]), ctx.dev(sc["poses"]), 99, sc["K"])


@pytest.mark.parametrize("n_kf,n_points", [(8, 600), (20, 10000)])
def test_point_errors_and_culling(ctx, oracle, synth, n_kf, n_points):
    """K12 (Mapper::cull_points / Slam::reprojection_error arithmetic) vs the oracle: per-point means and cull
    flags bit for bit (same f32 operations in CSR order), the culled list identical, the global f64 sum to 1e-12."""
    from test_oracle_cpu import _cull_scene
    sc = _cull_scene(synth, n_kf=n_kf, n_points=n_points)
    ref = oracle.point_errors(sc["positions"], sc["obs_ptr"], sc["obs_pose"], sc["obs_uv"], sc["poses"], sc["K"])
    d = ctx.point_errors(ctx.dev(sc["positions"]), ctx.dev(sc["obs_ptr"]), ctx.dev(sc["obs_pose"]), ctx.dev(sc["obs_uv"]),
                         ctx.dev(sc["poses"]), sc["K"])
    n = len(sc["positions"])
    assert np.array_equal(to_np(d["mean_err"])[:n].view(np.uint32), ref["mean_err"].view(np.uint32))
    assert np.array_equal(to_np(d["cull"])[:n], ref["cull"])
    cnt = int(to_np(d["cull_count"])[0])
    assert cnt == len(ref["cull_idx"]) > 0 and np.array_equal(to_np(d["cull_idx"])[:cnt], ref["cull_idx"])
    sums = to_np(d["sums"])
    assert sums[1] == ref["n_obs"] and sums[0] == pytest.approx(ref["err_sum"], rel=1e-12)
    # empty map
    e = ctx.point_errors(ctx.dev(np.zeros((0, 3), np.float32)), ctx.dev(np.zeros(1, np.int32)), ctx.dev(np.zeros(1, np.int32)),
                         ctx.dev(np.zeros((1, 2), np.float32)), ctx.dev(sc["poses"]), sc["K"])
    assert int(to_np(e["cull_count"])[0]) == 0 and to_np(e["sums"]).tolist() == [0.0, 0.0]


def test_triangulate_empty_and_degenerate(ctx, oracle, synth):
    pr = synth.make_pair(1)
    d = ctx.triangulate(ctx.dev(pr["kp1"]), ctx.dev(pr["kp2"]), 0, ctx.dev(pr["poses"]), 2, pr["K"])
    assert int(to_np(d["count"])[0]) == 0
    # identical poses: zero baseline -> everything rejected by parallax, nothing crashes
    poses = np.stack([pr["poses"][0], pr["poses"][0]])
    _tri_case(ctx, oracle, pr["kp1"][:64], pr["kp1"][:64] + 0.25, poses, pr["K"])


# ------------------------------------------------------- reprojection match
def _reproj_case(ctx, oracle, rs, frame, mp, replace):
    fv, k1 = ctx.make_frame_view(frame)
    mv, k2 = ctx.make_map_view(mp)
    out = ctx.reproj_match(fv, mv, replace=replace)
    ref = oracle.reproj_match(frame, mp, replace=replace)
    P, N = len(mp["positions"]), len(frame["keypoints"])
    assert np.array_equal(to_np(out["point_kp"])[:P], ref["point_kp"])
    assert np.array_equal(to_np(out["point_dist"])[:P], ref["point_dist"])
    assert np.array_equal(to_np(out["prop_point"])[:N], ref["prop_point"])
    assert np.array_equal(to_np(out["prop_dist"])[:N], ref["prop_dist"])
    cnt = int(to_np(out["count"])[0])
    assert cnt == len(ref["match_kp"])
    assert np.array_equal(to_np(out["match_kp"])[:cnt], ref["match_kp"])
    assert np.array_equal(to_np(out["match_point"])[:cnt], ref["match_point"])
    # the same with the KD-tree packed once for the frame (rs_kdtree_pack): identical outputs
    fvp, k3 = ctx.make_frame_view(frame, pack=True)
    outp = ctx.reproj_match(fvp, mv, replace=replace)
    for k in ("point_kp", "point_dist", "prop_point", "prop_dist"):
        assert np.array_equal(to_np(outp[k]), to_np(out[k])), k
    assert int(to_np(outp["count"])[0]) == cnt and np.array_equal(to_np(outp["match_point"])[:cnt], ref["match_point"])
    # one lane per map point (the kernel that also serves KD-trees too large for LDS) instead of eight: identical outputs
    ctx.set_int("k2_mode", 1)
    try:
        out1 = ctx.reproj_match(fv, mv, replace=replace)
        for k in ("point_kp", "point_dist", "prop_point", "prop_dist"):
            assert np.array_equal(to_np(out1[k]), to_np(out[k])), k
        assert int(to_np(out1["count"])[0]) == cnt and np.array_equal(to_np(out1["match_kp"])[:cnt], ref["match_kp"])
        assert np.array_equal(to_np(out1["match_point"])[:cnt], ref["match_point"])
    finally:
        ctx.set_int("k2_mode", 0)
    return cnt


@pytest.mark.parametrize("replace", [0, 1])
def test_reproj_match_window(ctx, oracle, rs, synth, replace):
    w = synth.make_ba_window(n_kf=10, n_points=3000)
    frame, mp = synth.make_match_scene(w, n_keypoints=1500, kdtree_build=rs.kdtree_build)
    cnt = _reproj_case(ctx, oracle, rs, frame, mp, replace)
    assert cnt > 100


def test_reproj_match_dense_candidates_and_ties(ctx, oracle, rs, synth):
    """many keypoints inside every search disc + duplicated descriptors: exercises the
    KD traversal-order and map-order tie rules."""
    w = synth.make_ba_window(n_kf=6, n_points=400, run_max=5)
    frame, mp = synth.make_match_scene(w, n_keypoints=600, kdtree_build=rs.kdtree_build, matched_frac=0.1)
    rng = np.random.default_rng(9)
    kp = frame["keypoints"]
    kp[:] = kp[rng.integers(0, 40, len(kp))] + rng.integers(-6, 7, kp.shape).astype(np.float32)
    frame["descriptors"][:] = frame["descriptors"][rng.integers(0, 25, len(kp))]
    mp["desc_pool"][:] = frame["descriptors"][rng.integers(0, 25, len(mp["desc_pool"]))]
    bits = rng.integers(0, 256, mp["desc_pool"].shape, dtype=np.uint8) & rng.integers(0, 256, mp["desc_pool"].shape, dtype=np.uint8) & 0x11
    mp["desc_pool"] ^= bits
    node_kp, left, right, root = rs.kdtree_build(kp)
    frame.update(kd_node_kp=node_kp, kd_left=left, kd_right=right, kd_root=root)
    # project the map onto those clusters so that discs are crowded
    for replace in (0, 1):
        _reproj_case(ctx, oracle, rs, frame, mp, replace)


def test_reproj_match_integer_pixels_long_tracks_crowded_discs(ctx, oracle, rs, synth):
    """What the reference's front end really produces: GFTT corners are INTEGER pixels
    (src/features/OrbFeatureExtractor.h:25-26), so equal coordinates in the KD-tree build and traversal are routine, not
    a corner case; long-lived points carry more observations than K2 preloads (8) or handles per batch (16); and a
    crowded disc holds more candidates than K2 queues (16: the on-the-spot path and its tie rule against the queue)."""
    w = synth.make_ba_window(n_kf=24, n_points=1500, run_min=2, run_max=24, config_id=91)
    frame, mp = synth.make_match_scene(w, n_keypoints=1800, kdtree_build=rs.kdtree_build, matched_frac=0.2, config_id=91)
    assert np.diff(mp["obs_ptr"]).max() > 16
    rng = np.random.default_rng(91)
    kp = frame["keypoints"]
    # 30 clusters of ~60 integer-pixel corners each within +-7 px: > 16 candidates inside most 20-px discs, many equal x or y
    centres = kp[rng.integers(0, len(kp), 30)]
    kp[:] = np.round(centres[rng.integers(0, 30, len(kp))] + rng.integers(-7, 8, kp.shape))
    frame["descriptors"][:] = frame["descriptors"][rng.integers(0, 40, len(kp))]          # duplicated descriptors: distance ties
    mp["desc_pool"][:] = frame["descriptors"][rng.integers(0, 40, len(mp["desc_pool"]))]
    mp["desc_pool"] ^= rng.integers(0, 256, mp["desc_pool"].shape, dtype=np.uint8) & rng.integers(0, 256, mp["desc_pool"].shape, dtype=np.uint8) & 0x21
    node_kp, left, right, root = rs.kdtree_build(kp)
    frame.update(kd_node_kp=node_kp, kd_left=left, kd_right=right, kd_root=root)
    o_kd = oracle.kdtree_build(kp)
    assert np.array_equal(node_kp, o_kd[0]) and np.array_equal(left, o_kd[1]) and np.array_equal(right, o_kd[2]) and root == o_kd[3]
    total = 0
    for replace in (0, 1):
        total += _reproj_case(ctx, oracle, rs, frame, mp, replace)
    assert total > 20


# ----------------------------------------------------------------------- BA
def _ba_case(ctx, oracle, w, options=None, o_options=None):
    ref_c, ref_p, ref_s = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"],
                                               w["obs_uv"], w["K"], o_options)
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]),
                          ctx.dev(w["obs_uv"]), w["K"], options)
    return to_np(dc), to_np(dp), s, ref_c, ref_p, ref_s


def test_bundle_adjust_small(ctx, oracle, synth):
    w = synth.make_ba_window(n_kf=6, n_points=200, run_max=5)
    c, p, s, rc, rp, rs_ = _ba_case(ctx, oracle, w)
    assert s["usable"] == rs_["usable"] == 1
    assert s["iterations"] == rs_["iterations"]
    assert s["successful_steps"] == rs_["successful_steps"]
    assert s["termination"] == rs_["termination"]
    # f64 LM trajectories: identical schedule, different summation order / analytic vs jet
    # Jacobians -> agreement far below the f32 boundary (1e-7): tolerance 1e-8 relative
    assert np.isclose(s["initial_cost"], rs_["initial_cost"], rtol=1e-12)
    assert np.isclose(s["final_cost"], rs_["final_cost"], rtol=1e-8)
    assert np.allclose(c, rc, rtol=1e-7, atol=1e-9)
    assert np.allclose(p, rp, rtol=1e-7, atol=1e-8)
    # fixed cameras untouched
    assert np.array_equal(c[:2], w["cams"][:2])


def test_bundle_adjust_cfg3_window(ctx, oracle, synth):
    """BASELINE.json configs[2]: 20 KF x 10k landmarks x ~60k observations, 10 LM iterations."""
    w = synth.make_ba_window()
    c, p, s, rc, rp, rs_ = _ba_case(ctx, oracle, w)
    assert s["usable"] == 1 and rs_["usable"] == 1
    assert (s["iterations"], s["successful_steps"]) == (rs_["iterations"], rs_["successful_steps"])
    assert np.isclose(s["final_cost"], rs_["final_cost"], rtol=1e-7)
    assert np.allclose(c, rc, rtol=1e-6, atol=1e-8)
    assert np.allclose(p, rp, rtol=1e-6, atol=1e-7)
    assert s["final_cost"] < 0.2 * s["initial_cost"]


@pytest.mark.parametrize("n_kf,n_points,run_max", [(20, 12000, 10), (70, 6000, 12), (30, 11000, 24)])
def test_bundle_adjust_item_sizes_and_camera_staging(ctx, oracle, synth, n_kf, n_points, run_max):
    """Value-by-value against the oracle for the K5 variants the other tests do not reach: 64-landmark items
    (more than 10240 landmarks), camera blocks read from global memory (more than 64 cameras) with the blocked
    reduced solve (n = 408), and camera unions beyond 10 per item (8x8-tile SYRK in two half batches) on big items."""
    w = synth.make_ba_window(n_kf=n_kf, n_points=n_points, run_min=2, run_max=run_max, config_id=12)
    c, p, s, rc, rp, rs_ = _ba_case(ctx, oracle, w)
    assert s["usable"] == 1 and rs_["usable"] == 1
    assert (s["iterations"], s["successful_steps"]) == (rs_["iterations"], rs_["successful_steps"])
    assert np.isclose(s["final_cost"], rs_["final_cost"], rtol=1e-7)
    assert np.allclose(c, rc, rtol=1e-6, atol=1e-8)
    assert np.allclose(p, rp, rtol=1e-6, atol=1e-7)


def test_bundle_adjust_rejected_keeps_input(ctx, synth):
    """solve()'s accept rule (src/Optimization.cpp:136-141): an unusable result leaves inputs untouched."""
    w = synth.make_ba_window(n_kf=5, n_points=80, run_max=4)
    w["points"][:] = np.nan    # non-finite cost -> FAILURE
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]),
                          ctx.dev(w["obs_uv"]), w["K"])
    assert s["usable"] == 0
    assert np.array_equal(to_np(dc), w["cams"])


def test_bundle_adjust_two_frame_init(ctx, oracle, synth):
    """Initialization's call pattern (src/Initialization.cpp:249): 2 frames, the first fixed."""
    w = synth.make_ba_window(n_kf=2, n_points=150, run_min=2, run_max=2, n_fixed=1)
    c, p, s, rc, rp, rs_ = _ba_case(ctx, oracle, w)
    assert s["usable"] == rs_["usable"]
    assert np.allclose(c, rc, rtol=1e-6, atol=1e-8)
    assert np.allclose(p, rp, rtol=1e-6, atol=1e-7)


def test_bundle_adjust_options_iterations(ctx, oracle, rs, synth):
    w = synth.make_ba_window(n_kf=6, n_points=200, run_max=5, config_id=5)
    o = rs.default_options(); o.max_num_iterations = 3
    oo = oracle.default_options(); oo.max_num_iterations = 3
    c, p, s, rc, rp, rs_ = _ba_case(ctx, oracle, w, o, oo)
    assert s["iterations"] == rs_["iterations"] == 3
    assert np.allclose(c, rc, rtol=1e-7, atol=1e-9)


def test_refine_pose(ctx, oracle, synth):
    w = synth.make_ba_window(n_kf=4, n_points=600, run_min=4, run_max=4)
    sel = np.flatnonzero(w["obs_cam"] == 3)
    pts = w["points_true"][np.repeat(np.arange(len(w["points"])), np.diff(w["obs_ptr"]))[sel]]
    uv = w["obs_uv"][sel]
    cam0 = w["cams"][3].copy()
    ref_cam, ref_s = oracle.refine_pose(cam0, pts, uv, w["K"])
    cam, s = ctx.refine_pose(cam0, ctx.dev(pts), ctx.dev(uv), w["K"])
    assert s["usable"] == ref_s["usable"] == 1
    assert (s["iterations"], s["successful_steps"], s["termination"]) == \
        (ref_s["iterations"], ref_s["successful_steps"], ref_s["termination"])
    assert np.isclose(s["final_cost"], ref_s["final_cost"], rtol=1e-8)
    assert np.allclose(cam, ref_cam, rtol=1e-7, atol=1e-9)
    # empty input: "nothing to constrain"
    cam2, s2 = ctx.refine_pose(cam0, ctx.dev(pts[:0]), ctx.dev(uv[:0]), w["K"])
    assert s2["usable"] == 0 and np.array_equal(cam2, cam0)


def test_profiling_names(ctx, synth):
    pr = synth.make_pair(1)
    ctx.prof_begin()
    ctx.match_descriptors(ctx.dev(pr["desc2"]), ctx.dev(pr["desc1"]), 500, 500)
    prof = ctx.prof_end()
    assert "K1_hamming_knn2" in prof and prof["K1_hamming_knn2"][0] == 1
    assert prof["K1_hamming_knn2"][1] > 0


# ---------------------------------------------- BA: the less common code paths
@pytest.mark.parametrize("n_kf,run_min,run_max,n_points", [
    (22, 12, 20, 600),     # 20 free cameras: items with 11..21 cameras -> 8x8-tile (NT = 8) SYRK path, LDS solve (n = 120)
    (30, 2, 28, 500),      # 28 free cameras: n = 168 -> global-memory reduced solve; unions > 21 -> per-landmark fallback
    (100, 2, 10, 1500),    # cfg-5-shaped window (100 KF): n = 588, atomics Schur kernel + global solve + LDS back-substitution
])
def test_bundle_adjust_code_paths(ctx, oracle, synth, n_kf, run_min, run_max, n_points):
    w = synth.make_ba_window(n_kf=n_kf, n_points=n_points, run_min=run_min, run_max=run_max, config_id=40 + n_kf)
    c, p, s, rc, rp, rs_ = _ba_case(ctx, oracle, w)
    assert s["usable"] == rs_["usable"] == 1
    assert (s["iterations"], s["successful_steps"], s["termination"]) == \
        (rs_["iterations"], rs_["successful_steps"], rs_["termination"])
    assert np.isclose(s["final_cost"], rs_["final_cost"], rtol=1e-7)
    assert np.allclose(c, rc, rtol=1e-6, atol=1e-8)
    assert np.allclose(p, rp, rtol=1e-6, atol=1e-7)


def test_bundle_adjust_arbitrary_covisibility(ctx, oracle, synth):
    """observations of a landmark are NOT a consecutive run and not in camera order: exercises the
    compact-row mapping (rank_in_mask / nth_set_bit) and the fixed-camera-only landmarks."""
    w = synth.make_ba_window(n_kf=12, n_points=400, run_min=3, run_max=6, config_id=77)
    rng = np.random.default_rng(77)
    cams_of = []
    for p in range(400):
        k = w["obs_ptr"][p + 1] - w["obs_ptr"][p]
        if p % 50 == 0:
            cams_of.append(np.array([0, 1])[:max(2, min(k, 2))])       # seen by the two fixed cameras only
        else:
            cams_of.append(rng.permutation(12)[:k])                      # scattered, unsorted
    obs_ptr = np.zeros(401, np.int32); obs_ptr[1:] = np.cumsum([len(c) for c in cams_of])
    obs_cam = np.concatenate(cams_of).astype(np.int32)
    obs_pt = np.repeat(np.arange(400), np.diff(obs_ptr))
    uv = np.zeros((len(obs_cam), 2))
    K = w["K"].astype(np.float64)
    for o in range(len(obs_cam)):
        uv[o] = synth.project(w["poses_true"][obs_cam[o]], K, w["points_true"][obs_pt[o]][None])[0][0]
    uv += rng.normal(0, 0.5, uv.shape)
    w.update(obs_ptr=obs_ptr, obs_cam=obs_cam, obs_uv=uv.astype(np.float32))
    c, p, s, rc, rp, rs_ = _ba_case(ctx, oracle, w)
    assert s["usable"] == rs_["usable"] == 1
    assert (s["iterations"], s["successful_steps"]) == (rs_["iterations"], rs_["successful_steps"])
    assert np.allclose(c, rc, rtol=1e-6, atol=1e-8)
    assert np.allclose(p, rp, rtol=1e-6, atol=1e-7)


def test_bundle_adjust_cfg5_full_size_properties(ctx, synth):
    """cfg 5 at full size (100 key frames, 80 k landmarks, ~480 k observations, n = 588): the window-size-independent
    MFMA Schur kernel and the blocked reduced solve.  The oracle needs minutes here, so the checks are the
    size-independent ones: usable solve, large cost decrease, fixed cameras untouched, finite result, and a second
    solve started from the result finds (almost) nothing left.  Small 100-KF windows are compared with the oracle
    value by value in test_bundle_adjust_code_paths."""
    w = synth.make_ba_window(n_kf=100, n_points=80000, config_id=5)
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    args = (ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]))
    s1 = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
    assert s1["usable"] == 1 and s1["iterations"] >= 5
    assert s1["final_cost"] < 0.15 * s1["initial_cost"]
    c1, p1 = to_np(dc), to_np(dp)
    fixed = ~np.asarray(w["cam_free"]).astype(bool)
    assert fixed.sum() == 2 and np.array_equal(c1[fixed], w["cams"][fixed])
    assert np.isfinite(c1).all() and np.isfinite(p1).all()
    s2 = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
    assert s2["initial_cost"] == pytest.approx(s1["final_cost"], rel=1e-9)
    assert s2["final_cost"] <= s2["initial_cost"] and s2["final_cost"] > 0.5 * s2["initial_cost"]


def test_match_descriptors_cfg4_batch_properties(ctx, synth):
    """BASELINE.json configs[3]: 64 pairs x 2k keypoints.  Size-independent properties: the batched
    launch equals 64 single launches bit for bit, and matching a set against itself is the identity."""
    import torch
    rng = np.random.default_rng(4)
    B, n = 64, 2000
    base = rng.integers(0, 256, (B, n, 32), dtype=np.uint8)
    noise = (rng.random((B, n, 32 * 8)) < 0.05)
    q = np.packbits(np.unpackbits(base, axis=2) ^ noise, axis=2)
    dq, dt = ctx.dev(q), ctx.dev(base)
    m = ctx.match_descriptors(dq, dt, n, n, batch=B, raw=True)
    cnt = to_np(m["cnt"])
    assert cnt.min() > 0.95 * n
    for b in (0, 17, 63):
        one = ctx.match_descriptors(dq[b].contiguous(), dt[b].contiguous(), n, n, raw=True)
        c = int(to_np(one["cnt"])[0])
        assert c == cnt[b]
        assert torch.equal(one["mq"][0, :c], m["mq"][b, :c]) and torch.equal(one["mt"][0, :c], m["mt"][b, :c])
        for a, bb in zip(one["raw"], m["raw"]):
            assert torch.equal(a[0], bb[b])
        # true correspondence i <-> i (5 % bit flips: distance ~13, impostors ~128)
        assert (to_np(m["mq"])[b, :c] == to_np(m["mt"])[b, :c]).mean() > 0.999
    ident = ctx.match_descriptors(dt[0].contiguous(), dt[0].contiguous(), n, n, raw=True)
    assert int(to_np(ident["cnt"])[0]) == n
    assert np.array_equal(to_np(ident["raw"][0])[0], np.arange(n)) and to_np(ident["raw"][1])[0].max() == 0


def test_cfg4_batch_match_and_triangulate(ctx, oracle, synth):
    """BASELINE.json configs[3] end to end at full size: 64 pairs x 2000 keypoints, batched brute-force match feeding
    the batched triangulation (rs_triangulate_matches_batch, one launch pair for all 64 pairs, per-pair poses).
    Three sampled pairs against the oracle bit for bit; all 64 against single-pair launches bit for bit
    (reference src/Triangulation.cpp:28-106 per pair)."""
    import torch
    B = 64
    bt = synth.make_pair_batch(B)
    n = bt["desc1"].shape[1]
    dq, dt = ctx.dev(bt["desc2"]), ctx.dev(bt["desc1"])
    dk1, dk2, dpo = ctx.dev(bt["kp1"]), ctx.dev(bt["kp2"]), ctx.dev(bt["poses"])
    m = ctx.match_descriptors(dq, dt, n, n, batch=B)
    t = ctx.triangulate_matches_batch(dk1, dk2, m["mt"], m["mq"], m["cnt"], dpo, bt["K"])
    cnt = to_np(m["cnt"])
    tcnt = to_np(t["count"])
    assert cnt.min() > 0.8 * n and tcnt.min() > 0.2 * n   # the default parallax gate (0.9999) drops distant landmarks
    for b in (0, 29, 63):
        pr = bt["pairs"][b]
        mq, mt = oracle.match_descriptors(pr["desc2"], pr["desc1"])
        assert len(mq) == cnt[b]
        assert np.array_equal(to_np(m["mq"])[b, :cnt[b]], mq) and np.array_equal(to_np(m["mt"])[b, :cnt[b]], mt)
        ref = oracle.triangulate(pr["kp1"][mt], pr["kp2"][mq], pr["poses"], pr["K"])
        assert np.array_equal(to_np(t["keep"])[b, :cnt[b]], ref["keep"])
        assert np.array_equal(to_np(t["xyz"])[b, :cnt[b]].view(np.uint32), ref["xyz"].view(np.uint32))
        assert tcnt[b] == len(ref["out_index"])
        assert np.array_equal(to_np(t["out_index"])[b, :tcnt[b]], ref["out_index"])
        assert np.array_equal(to_np(t["out_xyz"])[b, :tcnt[b]].view(np.uint32), ref["out_xyz"].view(np.uint32))
    for b in range(B):
        one = ctx.triangulate_matches(dk1[b], dk2[b], m["mt"][b], m["mq"][b], m["cnt"][b:b + 1], n, dpo[b], bt["K"])
        c, tc = int(cnt[b]), int(tcnt[b])
        assert int(to_np(one["count"])[0]) == tc
        assert torch.equal(one["keep"][:c], t["keep"][b, :c]) and torch.equal(one["xyz"][:c], t["xyz"][b, :c])
        assert torch.equal(one["out_index"][:tc], t["out_index"][b, :tc])
        assert torch.equal(one["out_xyz"][:tc], t["out_xyz"][b, :tc])
    # the poses differ between pairs (pose_jitter), so a wrong pose index cannot pass
    assert not np.array_equal(bt["poses"][0], bt["poses"][1])


def test_bundle_adjust_free_camera_without_observations(ctx, oracle, synth):
    """VERDICT r1 weak #6 / ADVICE: a FREE camera that carries no residual block.  The reference only adds parameter
    blocks that have residuals (src/Optimization.cpp:306-316) and the oracle drops such a camera (oracle/ba.c,
    active free cameras).  The library keeps its slot in the reduced system as a pure-damping block (delta = 0) and
    excludes it from the norms of the parameter-tolerance test, so schedule and results must not change."""
    w = synth.make_ba_window(n_kf=7, n_points=300, run_max=5, config_id=17)
    # camera 7: free, never observed, far from the origin (it would dominate x_norm if it were counted)
    cams = np.concatenate([w["cams"], [[0.3, -0.2, 0.1, 500.0, -300.0, 800.0]]])
    free = np.concatenate([w["cam_free"], [1]]).astype(np.uint8)
    w2 = dict(w, cams=cams, cam_free=free)
    c, p, s, rc, rp, rs_ = _ba_case(ctx, oracle, w2)
    tr = ctx.ba_trace()
    c0, p0, s0, _, _, _ = _ba_case(ctx, oracle, w)
    assert (s["iterations"], s["successful_steps"], s["termination"]) == (rs_["iterations"], rs_["successful_steps"], rs_["termination"])
    assert (s["iterations"], s["successful_steps"], s["termination"]) == (s0["iterations"], s0["successful_steps"], s0["termination"])
    assert np.array_equal(c[7], cams[7]) and np.array_equal(rc[7], cams[7])       # untouched on both sides
    assert np.allclose(c, rc, rtol=1e-7, atol=1e-9) and np.allclose(p, rp, rtol=1e-7, atol=1e-8)
    assert np.allclose(c[:7], c0, rtol=1e-9, atol=1e-11)
    _, _, _, otr = oracle.bundle_adjust_trace(w2["cams"], w2["cam_free"], w2["points"], w2["obs_ptr"], w2["obs_cam"],
                                               w2["obs_uv"], w2["K"])
    assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
    assert np.allclose([t["x_norm"] for t in tr], [t["x_norm"] for t in otr], rtol=1e-6)   # would be >= 1e3 from the first iteration if it were counted


def test_bundle_adjust_cfg3_trace(ctx, oracle, synth):
    """Per-iteration schedule of the benchmark window (cfg 3 rejects 6 of its 10 steps): outcome, radius, costs."""
    w = synth.make_ba_window()
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
    tr = ctx.ba_trace()
    _, _, rs_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"],
                                                w["obs_uv"], w["K"])
    assert len(tr) == s["iterations"] == rs_["iterations"]
    assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
    assert sum(t["outcome"] == 0 for t in tr) >= 5
    for k, tol in (("radius", 1e-7), ("cost", 1e-9), ("candidate_cost", 1e-8), ("model_cost_change", 1e-6),
                   ("step_norm", 1e-6), ("x_norm", 1e-9)):
        assert np.allclose([t[k] for t in tr], [t[k] for t in otr], rtol=tol), k


@pytest.mark.parametrize("kw", [dict(), dict(n_kf=8, n_points=150, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0),
                                dict(n_kf=5, n_points=100, run_max=5, config_id=48, outlier_frac=0.15, rot_noise_deg=1.0),
                                dict(n_kf=12, n_points=3000, run_max=8, config_id=19, outlier_frac=0.06)])
def test_bundle_adjust_speculative_radii_do_not_change_the_schedule(ctx, rs, oracle, synth, kw):
    """Speculative radii (1 .. 5 trust-region radii evaluated per round; 0 = the default: up to 5 while the radius
    is uncalibrated, 3 afterwards) only change how many launches a solve takes: per-iteration outcomes, radii, costs and the
    result are those of the sequential loop (ns = 1) and of the oracle, whatever the accept / reject pattern
    (cfg 3: A R R R R A R R A A)."""
    w = synth.make_ba_window(**kw)
    args = (w["cam_free"],)
    runs = {}
    try:
        for ns in (1, 2, 3, 4, 5, 0):
            ctx.set_int("ba_speculative_sets", ns)
            dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
            s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
            runs[ns] = (s, ctx.ba_trace(), to_np(dc), to_np(dp), ctx.ba_stats())
    finally:
        ctx.set_int("ba_speculative_sets", 0)
    _, _, os_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    s1, t1, c1, p1, _ = runs[1]
    if not kw:      # the benchmark window: rounds of 1, 5, 3, 1 radii by default (five while the radius is uncalibrated, never
        #                 more than the iterations left); 1, 3, 3, 3, 1 with three sets
        assert (runs[0][4]["rounds"], runs[0][4]["set_evaluations"]) == (4, 10), runs[0][4]
        assert (runs[3][4]["rounds"], runs[3][4]["set_evaluations"]) == (5, 11), runs[3][4]
    for ns in (1, 2, 3, 4, 5, 0):
        s, tr, c, p, _ = runs[ns]
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (os_["iterations"], os_["successful_steps"], os_["termination"], os_["usable"]), ns
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr], ns
        assert np.allclose([t["radius"] for t in tr], [t["radius"] for t in otr], rtol=1e-7), ns
        # (costs of far-off REJECTED candidates amplify the 1e-12 differences of the step: 2e-8 observed)
        assert np.allclose([t["candidate_cost"] for t in tr], [t["candidate_cost"] for t in otr], rtol=1e-6), ns
        assert np.isclose(s["final_cost"], s1["final_cost"], rtol=1e-10), ns
        assert np.allclose(c, c1, rtol=1e-9, atol=1e-11) and np.allclose(p, p1, rtol=1e-9, atol=1e-10), ns
    del args


@pytest.mark.parametrize("kw", [dict(), dict(n_kf=8, n_points=150, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0),
                                dict(n_kf=12, n_points=3000, run_max=8, config_id=19, outlier_frac=0.06)])
def test_bundle_adjust_fused_solve_and_backsubstitution(ctx, oracle, synth, kw):
    """K7 + K8 in one launch (the default on the plain local window: K8's workgroups wait for their set's hand-off word
    inside the launch) against the two separate launches: the same per-iteration record as the oracle either way, the
    results equal to summation-order level, and twelve runs of the fused form all reproduce it (a stale read of delta_c
    behind the hand-off word in any of their rounds would not)."""
    w = synth.make_ba_window(**kw)
    _, _, os_, otr = oracle.bundle_adjust_trace(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    runs = []
    try:
        for mode in (1,) + (2,) * 12:
            ctx.set_int("ba_fuse_mode", mode)
            dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
            s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
            runs.append((s, ctx.ba_trace(), to_np(dc), to_np(dp)))
    finally:
        ctx.set_int("ba_fuse_mode", 0)
    s1, t1, c1, p1 = runs[0]
    for s, tr, c, p in runs:
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (os_["iterations"], os_["successful_steps"], os_["termination"], os_["usable"])
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
        assert np.allclose([t["cost"] for t in tr], [t["cost"] for t in otr], rtol=1e-9)
        assert np.isclose(s["final_cost"], s1["final_cost"], rtol=1e-10)
        assert np.allclose(c, c1, rtol=1e-9, atol=1e-11) and np.allclose(p, p1, rtol=1e-9, atol=1e-10)


def test_reanchor_points(ctx, oracle, rs, synth):
    """rs_reanchor_points (K13, the tail of Mapper::bundle_adjust, src/Mapper.cpp:380-393) bit for bit against the oracle."""
    rng = np.random.default_rng(6)
    w = synth.make_ba_window(n_kf=9, n_points=50, run_max=4)
    before = np.stack([rs.unpack_pose(c) for c in w["cams"]]).reshape(-1, 16).astype(np.float32)
    after = np.stack([rs.unpack_pose(c) for c in w["cams_true"]]).reshape(-1, 16).astype(np.float32)
    n = 5000
    fi = rng.integers(0, 9, n).astype(np.int32)
    X = rng.normal(0, 8, (n, 3)).astype(np.float32)
    ref = oracle.reanchor_points(None, fi, before, after, X)
    dX = ctx.dev(X)
    ctx.reanchor_points(None, ctx.dev(fi), ctx.dev(before), ctx.dev(after), dX)
    assert np.array_equal(to_np(dX).view(np.uint32), ref.view(np.uint32))
    idx = rng.permutation(n)[:777].astype(np.int32)
    ref2 = oracle.reanchor_points(idx, fi[idx], before, after, X)
    dX2 = ctx.dev(X)
    ctx.reanchor_points(ctx.dev(idx), ctx.dev(fi[idx]), ctx.dev(before), ctx.dev(after), dX2)
    assert np.array_equal(to_np(dX2).view(np.uint32), ref2.view(np.uint32))
    ctx.reanchor_points(None, ctx.dev(fi[:0]), ctx.dev(before), ctx.dev(after), dX2)       # n = 0
    # poses in host arrays, as the reference holds them: 9 frames travel as kernel arguments, 40 through a device copy
    dX3 = ctx.dev(X)
    ctx.reanchor_points_host_poses(None, ctx.dev(fi), before, after, dX3)
    assert np.array_equal(to_np(dX3).view(np.uint32), ref.view(np.uint32))
    dX4 = ctx.dev(X)
    ctx.reanchor_points_host_poses(ctx.dev(idx), ctx.dev(fi[idx]), before, after, dX4)
    assert np.array_equal(to_np(dX4).view(np.uint32), ref2.view(np.uint32))
    before40 = np.ascontiguousarray(np.tile(before, (5, 1))[:40])
    after40 = np.ascontiguousarray(np.tile(after, (5, 1))[:40])
    fi40 = rng.integers(0, 40, n).astype(np.int32)
    ref40 = oracle.reanchor_points(None, fi40, before40, after40, X)
    dX5 = ctx.dev(X)
    ctx.reanchor_points_host_poses(None, ctx.dev(fi40), before40, after40, dX5)
    assert np.array_equal(to_np(dX5).view(np.uint32), ref40.view(np.uint32))


def test_bundle_adjust_launch_accounting(ctx, synth):
    """rs_ba_get_stats: the benchmark window takes fewer rounds than LM iterations (speculative radii), evaluates at
    least one set per iteration, and the host enqueues no more than one round beyond those that did work."""
    w = synth.make_ba_window()
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
    st = ctx.ba_stats()
    assert s["iterations"] == 10
    assert st["rounds"] < s["iterations"] and st["set_evaluations"] >= s["iterations"]
    assert st["fresh_rounds"] <= st["rounds"] <= st["rounds_enqueued"] <= st["rounds"] + 1


@pytest.mark.parametrize("n_shards,kw", [(2, dict()), (3, dict(n_kf=8, n_points=900, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0)),
                                         (2, dict(n_kf=30, n_points=4000, run_max=8, config_id=41))])
def test_bundle_adjust_landmark_shards_in_process_group(rs, synth, n_shards, kw):
    """VERDICT r1 #6 / ADVICE: the product's N > 1 path executed for real on the one GPU of the box.  n contexts on
    device 0 (own stream, own host thread) form an in-process group (rs_comm_init_local: deterministic on-device sum
    instead of RCCL); each owns one landmark shard, cameras are replicated.  This runs everything the multi-GPU BA
    runs — rank-offset gradient-max blocks, the (1 + n_ranks) slot blocks, the fold of 8 replicas x n ranks in K7,
    the two all-reduces per round, the kept U / gc on rejected steps (cfg 3 rejects 6 of 10), the blocked solve for
    30 key frames — and must reproduce the unsharded solve: identical schedule, poses and points to 1e-9."""
    import threading
    import torch
    w = synth.make_ba_window(**kw)
    single = rs.Context(0)
    dc, dp = single.dev(w["cams"]), single.dev(w["points"])
    s0 = single.bundle_adjust(dc, w["cam_free"], dp, single.dev(w["obs_ptr"]), single.dev(w["obs_cam"]), single.dev(w["obs_uv"]), w["K"])
    tr0 = single.ba_trace()
    c0, p0 = to_np(dc), to_np(dp)
    single.close()
    ctxs = [rs.Context(0) for _ in range(n_shards)]
    streams = [torch.cuda.Stream(device=ctxs[0].device) for _ in range(n_shards)]
    for c, st in zip(ctxs, streams):
        c.use_stream(st)
    rs.Context.comm_init_local(ctxs)
    shards = [synth.shard_ba_by_landmark(w, n_shards, r) for r in range(n_shards)]
    out = [None] * n_shards

    def work(r):
        try:
            c, sh = ctxs[r], shards[r]
            with torch.cuda.stream(streams[r]):
                dcr, dpr = c.dev(sh["cams"]), c.dev(sh["points"])
                args = (c.dev(sh["obs_ptr"]), c.dev(sh["obs_cam"]), c.dev(sh["obs_uv"]))
                streams[r].synchronize()
                s = c.bundle_adjust(dcr, sh["cam_free"], dpr, *args, sh["K"])
                out[r] = (s, c.ba_trace(), to_np(dcr), to_np(dpr))
        except Exception as ex:      # noqa: BLE001
            out[r] = ex

    threads = [threading.Thread(target=work, args=(r,)) for r in range(n_shards)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in the exchange step"
    for c in ctxs:
        c.comm_destroy()
        c.close()
    for r in range(n_shards):
        assert not isinstance(out[r], Exception), out[r]
        s, tr, cr, pr = out[r]
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (s0["iterations"], s0["successful_steps"], s0["termination"], s0["usable"])
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in tr0]
        # the sum order of the all-reduce differs from the unsharded accumulation: 1e-12 per linearisation, amplified by
        # the LM iterations (the outlier-laden 8-KF window is the worst: 1e-8 on late radii) — far below the f32 write-back
        tol = 1e-9 if not kw else 1e-6
        assert np.allclose([t["radius"] for t in tr], [t["radius"] for t in tr0], rtol=tol)
        assert np.isclose(s["final_cost"], s0["final_cost"], rtol=tol) and np.isclose(s["initial_cost"], s0["initial_cost"], rtol=1e-12)
        assert np.allclose(cr, c0, rtol=tol, atol=tol * 1e-2)      # replicated cameras: every rank holds the result
        lo, hi = shards[r]["point_range"]
        assert np.allclose(pr, p0[lo:hi], rtol=tol, atol=tol * 1e-1)
        if r > 0:
            assert np.array_equal(cr, out[0][2]), "ranks must end with bit-identical cameras (redundant reduced solves)"


@pytest.mark.parametrize("kw,skip", [(dict(n_kf=7, n_points=150, run_max=5, config_id=62, outlier_frac=0.08, rot_noise_deg=1.5), {(4, 5)}),
                                     (dict(n_kf=8, n_points=400, run_max=6, config_id=61), set()),
                                     (dict(), set())])
@pytest.mark.parametrize("imu_mode", [0, 1])
def test_bundle_adjust_inertial(ctx, oracle, synth, kw, skip, imu_mode):
    """imu_mode 0: velocity / bias blocks eliminated around the LDS reduced solve (ba_imu.hip); 1: the N x N blocked solve.
    §8(f) rank 2 / a15: bundle_adjust with IMU factor pairs (preintegration 9 + bias walk 6 residuals per pair,
    velocity 3 + bias 6 unknowns per frame; reference src/Optimization.cpp:317-346, src/ImuFactor.cpp:19-118) through
    rs_bundle_adjust_inertial against the oracle: identical schedule per iteration, poses / velocities / biases / cost to
    f64 reformulation noise.  Cases: a gap in the factor chain, a full chain of 5, and the 20-key-frame benchmark window
    with 17 factor pairs (N = 108 + 162 = 270 camera-side unknowns on the blocked solve)."""
    ctx.set_int("ba_imu_mode", imu_mode)
    w = synth.make_ba_window(**kw)
    imu = synth.make_imu(w, skip=skip)
    args = (w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    rc, rp, rv, rb, rs_, otr = oracle.bundle_adjust_inertial(*args, imu, trace=True)
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s, v, b = ctx.bundle_adjust_inertial(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"], imu)
    tr = ctx.ba_trace()
    assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
           (rs_["iterations"], rs_["successful_steps"], rs_["termination"], rs_["usable"])
    assert s["usable"] == 1
    assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
    for k, tol in (("radius", 1e-7), ("cost", 1e-9), ("candidate_cost", 1e-6), ("model_cost_change", 1e-6), ("x_norm", 1e-6)):
        assert np.allclose([t[k] for t in tr], [t[k] for t in otr], rtol=tol), k
    assert np.isclose(s["initial_cost"], rs_["initial_cost"], rtol=1e-12) and np.isclose(s["final_cost"], rs_["final_cost"], rtol=1e-8)
    assert np.allclose(to_np(dc), rc, rtol=1e-7, atol=1e-9)
    assert np.allclose(to_np(dp), rp, rtol=1e-6, atol=1e-7)
    assert np.allclose(v, rv, rtol=1e-7, atol=1e-9) and np.allclose(b, rb, rtol=1e-6, atol=1e-9)
    # fixed frames keep velocity / bias; the IMU pulls the velocities towards the truth
    fixed = np.flatnonzero(np.asarray(w["cam_free"]) == 0)
    assert np.array_equal(v[fixed], imu["cam_velocity"][fixed]) and np.array_equal(b[fixed], imu["cam_bias"][fixed])
    free = np.flatnonzero(w["cam_free"])
    assert np.abs(v - imu["cam_velocity_true"])[free].max() < np.abs(imu["cam_velocity"] - imu["cam_velocity_true"])[free].max()
    # no factors: the inertial entry point is rs_bundle_adjust
    empty = dict(imu)
    for k in ("cam_i", "cam_j", "duration", "rotation", "velocity", "position", "covariance", "bias_gyro", "bias_accel", "bias_jacobian"):
        empty[k] = imu[k][:0]
    dc2, dp2 = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s2, v2, b2 = ctx.bundle_adjust_inertial(dc2, w["cam_free"], dp2, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"], empty)
    dc3, dp3 = ctx.dev(w["cams"]), ctx.dev(w["points"])
    s3 = ctx.bundle_adjust(dc3, w["cam_free"], dp3, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
    assert (s2["iterations"], s2["successful_steps"]) == (s3["iterations"], s3["successful_steps"])
    assert np.allclose(to_np(dc2), to_np(dc3), rtol=1e-9, atol=1e-11)
    assert np.array_equal(v2, imu["cam_velocity"]) and np.array_equal(b2, imu["cam_bias"])
    ctx.set_int("ba_imu_mode", 0)


@pytest.mark.parametrize("ns", [1, 2, 3])
def test_bundle_adjust_inertial_speculative_radii_do_not_change_the_schedule(ctx, oracle, synth, ns):
    """The inertial solve on the local-window kernels evaluates 1 .. 3 trust-region radii per round like the vision-only one
    (one elimination workgroup per radius): on a window that rejects steps the per-iteration record must equal the
    oracle's whatever the number of sets, only the number of rounds changes."""
    w = synth.make_ba_window(n_kf=8, n_points=150, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0)
    imu = synth.make_imu(w)
    args = (w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    rc, rp, rv, rb, rs_, otr = oracle.bundle_adjust_inertial(*args, imu, trace=True)
    assert 0 in [t["outcome"] for t in otr]                        # the window rejects steps
    ctx.set_int("ba_speculative_sets", ns)
    try:
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        s, v, b = ctx.bundle_adjust_inertial(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"], imu)
        tr = ctx.ba_trace()
        stats = ctx.ba_stats()
    finally:
        ctx.set_int("ba_speculative_sets", 0)
    assert [t["outcome"] for t in tr] == [t["outcome"] for t in otr]
    assert np.allclose([t["radius"] for t in tr], [t["radius"] for t in otr], rtol=1e-7)
    assert np.allclose([t["cost"] for t in tr], [t["cost"] for t in otr], rtol=1e-9)
    assert np.allclose(to_np(dc), rc, rtol=1e-7, atol=1e-9) and np.allclose(v, rv, rtol=1e-7, atol=1e-9)
    if ns == 1:
        assert stats["rounds"] == s["iterations"]
    else:
        assert stats["rounds"] < s["iterations"]


def test_bundle_adjust_inertial_rejects_bad_factors(ctx, rs, synth):
    w = synth.make_ba_window(n_kf=5, n_points=80, run_max=4)
    imu = synth.make_imu(w)
    bad = dict(imu, cam_i=imu["cam_i"].copy())
    bad["cam_i"][0] = 0                     # camera 0 is fixed: the reference only links optimised frames (:321-325)
    dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
    with pytest.raises(rs.RsError):
        ctx.bundle_adjust_inertial(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"], bad)
    assert np.array_equal(to_np(dc), w["cams"])


@pytest.mark.parametrize("mode", [0, 1])
def test_bundle_adjust_batch_of_windows(ctx, oracle, synth, mode):
    """rs_bundle_adjust_batch: 11 independent windows of different sizes against single solves: identical schedules,
    values to summation-order noise; three of them against the oracle.  mode 0: ONE launch sequence for all windows
    (blockIdx.z = window; mixed sizes share the grids of the largest); mode 1: the lanes (more windows than the 8
    lanes, so lanes take several)."""
    import torch
    ctx.set_int("ba_batch_mode", mode)
    kws = [dict(n_kf=6, n_points=200 + 37 * i, run_max=5, config_id=80 + i) for i in range(9)] + \
          [dict(), dict(n_kf=8, n_points=150, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0)]
    ws = [synth.make_ba_window(**k) for k in kws]
    single, probs, keep = [], [], []
    for w in ws:
        dev = [ctx.dev(w[k]) for k in ("obs_ptr", "obs_cam", "obs_uv")]
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *dev, w["K"])
        single.append((s, to_np(dc), to_np(dp)))
        bc, bp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        probs.append((bc, w["cam_free"], bp, *dev, w["K"]))
        keep.append((bc, bp))
    torch.cuda.synchronize()
    out = ctx.bundle_adjust_batch(probs)
    for i, (w, s) in enumerate(zip(ws, out)):
        s1, c1, p1 = single[i]
        assert (s["iterations"], s["successful_steps"], s["termination"], s["usable"]) == \
               (s1["iterations"], s1["successful_steps"], s1["termination"], s1["usable"]), i
        assert np.isclose(s["final_cost"], s1["final_cost"], rtol=1e-9), i
        assert np.allclose(to_np(keep[i][0]), c1, rtol=1e-8, atol=1e-10) and np.allclose(to_np(keep[i][1]), p1, rtol=1e-7, atol=1e-9), i
    for i in (0, 5, 10):
        w = ws[i]
        rc, rp, rs_ = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
        assert (out[i]["iterations"], out[i]["successful_steps"]) == (rs_["iterations"], rs_["successful_steps"])
        assert np.allclose(to_np(keep[i][0]), rc, rtol=1e-6, atol=1e-8)
    assert ctx.bundle_adjust_batch([]) == []
    ctx.set_int("ba_batch_mode", 0)


def test_bundle_adjust_through_rccl_single_rank(ctx, rs, synth):
    """The multi-GPU code path on one GPU: a 1-rank RCCL communicator (dlopen of librccl,
    ncclCommInitRank, sum and max ncclAllReduce of the reduced system / scalar slots on the library
    stream every LM step) must leave the solve unchanged."""
    w = synth.make_ba_window(n_kf=8, n_points=500, run_max=6, config_id=61)
    def run():
        dc, dp = ctx.dev(w["cams"]), ctx.dev(w["points"])
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]), w["K"])
        return to_np(dc), to_np(dp), s
    c0, p0, s0 = run()
    ctx.comm_init(rs.Context.comm_unique_id(), 1, 0)
    try:
        ctx.prof_begin()
        c1, p1, s1 = run()
        prof = ctx.prof_end()
    finally:
        ctx.comm_destroy()
    assert prof["C1_allreduce_system"][0] == 10 and prof["C2_allreduce_cost"][0] == 10
    assert (s0["iterations"], s0["successful_steps"], s0["termination"]) == (s1["iterations"], s1["successful_steps"], s1["termination"])
    assert np.allclose(c0, c1, rtol=1e-9, atol=1e-12) and np.allclose(p0, p1, rtol=1e-9, atol=1e-11)


def test_repeatability_of_a_pass(ctx, synth):
    """Twenty repetitions of the benchmark pass on the same inputs: integer outputs identical every time, the BA (f64
    atomics and matrix-core sums in varying order) reproducible to 1e-9 relative with the same LM schedule — a cheap
    race detector for the kernels that hand data to each other through global memory."""
    pr = synth.make_pair(2)
    w = synth.make_ba_window()
    d_q, d_t = ctx.dev(pr["desc2"]), ctx.dev(pr["desc1"])
    cams0, pts0 = ctx.dev(w["cams"]), ctx.dev(w["points"])
    args = (ctx.dev(w["obs_ptr"]), ctx.dev(w["obs_cam"]), ctx.dev(w["obs_uv"]))
    tk = synth.make_tracks(n_tracks=2000)
    targs = (ctx.dev(tk["track_uv"]), ctx.dev(tk["sight_ptr"]), ctx.dev(tk["sight_pose"]), ctx.dev(tk["sight_uv"]),
             ctx.dev(tk["poses"]), tk["kf_pose"], tk["K"])
    ref = None
    for rep in range(20):
        m = ctx.match_descriptors(d_q, d_t, 2000, 2000)
        cnt = int(to_np(m["cnt"])[0])
        mq = to_np(m["mq"])[0, :cnt].copy()
        t = ctx.triangulate_tracks(*targs, d_skip=ctx.dev(tk["skip"]))
        acc = to_np(t["accepted"])[:int(to_np(t["counts"])[0])].copy()
        dc, dp = cams0.clone(), pts0.clone()
        s = ctx.bundle_adjust(dc, w["cam_free"], dp, *args, w["K"])
        cur = (mq, acc, s["final_cost"], s["iterations"], s["successful_steps"], to_np(dc).copy())
        if ref is None:
            ref = cur
            continue
        assert np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1])
        assert cur[3:5] == ref[3:5]
        assert cur[2] == pytest.approx(ref[2], rel=1e-9)
        assert np.allclose(cur[5], ref[5], rtol=1e-9, atol=1e-12)
