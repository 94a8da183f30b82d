"""CPU tests (no GPU): the oracle against independent numpy/scipy formulations.

The reference ships no fixtures for this path (PARITY UNPINNED), so the
restatement in oracle/*.c is cross-checked here against formulations that share
no code with it: numpy bit counting, numpy.linalg.svd, finite differences, a
dense (non-Schur) numpy Levenberg-Marquardt step and scipy's robust least
squares.
"""
import numpy as np
import pytest


def popcount_rows(a, b):
    x = np.bitwise_xor(a[:, None, :], b[None, :, :])
    return np.unpackbits(x, axis=2).sum(axis=2).astype(np.int32)


# ------------------------------------------------------------------ matching
@pytest.mark.parametrize("nq,nt", [(1, 1), (5, 1), (33, 2), (200, 150), (64, 700)])
def test_knn2_vs_numpy(oracle, nq, nt):
    rng = np.random.default_rng(nq + 13 * nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    t[nt // 2] = t[0]                      # force an exact tie
    D = popcount_rows(q, t)
    order = np.argsort(D, axis=1, kind="stable")   # stable => lower train index first on ties
    i0, d0, i1, d1 = oracle.hamming_knn2(q, t)
    assert np.array_equal(i0, order[:, 0])
    assert np.array_equal(d0, D[np.arange(nq), order[:, 0]])
    if nt >= 2:
        assert np.array_equal(i1, order[:, 1])
        assert np.array_equal(d1, D[np.arange(nq), order[:, 1]])
    else:
        assert np.all(i1 == -1) and np.all(d1 == -1)


def test_match_descriptors_filters(oracle, synth):
    pr = synth.make_pair(1)
    q, t = pr["desc2"], pr["desc1"]
    D = popcount_rows(q, t)
    order = np.argsort(D, axis=1, kind="stable")
    d0 = D[np.arange(len(q)), order[:, 0]]
    d1 = D[np.arange(len(q)), order[:, 1]]
    # reference src/MapMatcher.cpp:152,156 in floats: reject d0 > 64, reject d0 > 0.75f * d1
    keep = ~(d0.astype(np.float32) > np.float32(64)) & ~(d0.astype(np.float32) > np.float32(0.75) * d1.astype(np.float32))
    mq, mt = oracle.match_descriptors(q, t)
    assert np.array_equal(mq, np.flatnonzero(keep))
    assert np.array_equal(mt, order[keep, 0])
    # ground truth: (almost) every accepted pair is a true correspondence
    a1, a2, ns = pr["truth12"]
    inv1 = np.empty_like(a1); inv1[a1] = np.arange(len(a1))
    inv2 = np.empty_like(a2); inv2[a2] = np.arange(len(a2))
    assert np.mean(inv2[mq] == inv1[mt]) > 0.99


def test_match_descriptors_edge_cases(oracle):
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    assert len(oracle.match_descriptors(q, q[:0])[0]) == 0           # empty train (:139-141)
    assert len(oracle.match_descriptors(q[:0], q)[0]) == 0           # empty query
    mq, mt = oracle.match_descriptors(q, q[:1])                      # k = 1: no ratio test (:145)
    assert np.array_equal(mq, [0]) and np.array_equal(mt, [0])
    # threshold boundaries: d0 == 64 accepted, 65 rejected; 4*d0 == 3*d1 accepted
    base = np.zeros((1, 32), np.uint8)
    def with_bits(n):
        b = np.zeros(256, np.uint8); b[:n] = 1
        return np.packbits(b)[None, :]
    t = np.concatenate([with_bits(64), with_bits(200)])
    assert len(oracle.match_descriptors(base, t)[0]) == 1
    t = np.concatenate([with_bits(65), with_bits(200)])
    assert len(oracle.match_descriptors(base, t)[0]) == 0
    t = np.concatenate([with_bits(48), with_bits(64)])               # 4*48 == 3*64
    assert len(oracle.match_descriptors(base, t)[0]) == 1
    t = np.concatenate([with_bits(49), with_bits(64)])
    assert len(oracle.match_descriptors(base, t)[0]) == 0


# ------------------------------------------------------------------- KD tree
def test_kdtree_radius_is_exact_and_ordered(oracle):
    rng = np.random.default_rng(2)
    kp = np.round(rng.uniform(0, 200, (500, 2))).astype(np.float32)   # integer pixels: many coordinate ties
    tree = oracle.kdtree_build(kp)
    node_kp, left, right, root = tree
    assert sorted(node_kp.tolist()) == list(range(500))
    for _ in range(50):
        x, y = rng.uniform(0, 200, 2)
        got = oracle.kdtree_radius(kp, tree, x, y, 20.0)
        d2 = (kp[:, 0] - np.float32(x)) ** 2 + (kp[:, 1] - np.float32(y)) ** 2
        assert sorted(got.tolist()) == np.flatnonzero(d2 <= np.float32(400.0)).tolist()

    # traversal order restated independently (recursive python, src/KDTree.cpp:52-82)
    def walk(node, depth, x, y, out):
        if node < 0:
            return
        k = node_kp[node]
        dx, dy = kp[k, 0] - np.float32(x), kp[k, 1] - np.float32(y)
        if dx * dx + dy * dy <= np.float32(400.0):
            out.append(k)
        delta = dx if depth % 2 == 0 else dy
        near, far = (left[node], right[node]) if delta > 0 else (right[node], left[node])
        walk(near, depth + 1, x, y, out)
        if delta * delta <= np.float32(400.0):
            walk(far, depth + 1, x, y, out)
    out = []
    walk(root, 0, 77.3, 101.9, out)
    assert oracle.kdtree_radius(kp, tree, 77.3, 101.9, 20.0).tolist() == out


def test_kdtree_host_library_matches_oracle(oracle, rs):
    """rs_kdtree_build (librsgpu host code, std::nth_element with a total order) builds the same tree."""
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 17, 500):
        kp = np.round(rng.uniform(0, 64, (n, 2))).astype(np.float32)
        a = rs.kdtree_build(kp)
        b = oracle.kdtree_build(kp)
        assert a[3] == b[3] or n == 0
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(x, y)


# ------------------------------------------------------- reprojection match
def test_reproj_match_against_python_restatement(oracle, synth):
    w = synth.make_ba_window(n_kf=5, n_points=150, run_max=4)
    frame, mp = synth.make_match_scene(w, n_keypoints=300, kdtree_build=oracle.kdtree_build)
    out = oracle.reproj_match(frame, mp, replace=0)
    # independent float64 restatement of the gates + brute-force radius search; integer outputs
    # must agree except for points sitting within float rounding of a gate (none at this seed)
    T = frame["pose"].reshape(4, 4).astype(np.float64)
    K = frame["K"].astype(np.float64)
    centre = -T[:3, :3].T @ T[:3, 3]
    N = len(frame["keypoints"])
    prop_d = np.full(N, 64); prop_p = np.full(N, -1)
    for p in range(len(mp["positions"])):
        if not mp["eligible"][p]:
            continue
        X = mp["positions"][p].astype(np.float64)
        pc = T[:3, :3] @ X + T[:3, 3]
        if pc[2] < 0:
            continue
        u, v = K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]
        if not (0 <= u < frame["width"] and 0 <= v < frame["height"]):
            continue
        obs = range(mp["obs_ptr"][p], mp["obs_ptr"][p + 1])
        dirs = [X - mp["kf_centers"][mp["obs_kf"][o]] for o in obs]
        dist = [np.linalg.norm(d) for d in dirs]
        normal = sum(d / np.linalg.norm(d) for d in dirs)
        normal /= np.linalg.norm(normal)
        ray = X - centre
        if normal @ (ray / np.linalg.norm(ray)) < 0.5:
            continue
        if np.linalg.norm(ray) < min(dist) / 2 or np.linalg.norm(ray) > max(dist) * 1.25:
            continue
        cand = oracle.kdtree_radius(frame["keypoints"], (frame["kd_node_kp"], frame["kd_left"], frame["kd_right"],
                                                         frame["kd_root"]), u, v, 20.0)
        best_d, best_k = 64, 0
        for k in cand:
            if frame["kp_matched"][k]:
                continue
            for o in obs:
                d = int(np.unpackbits(frame["descriptors"][k] ^ mp["desc_pool"][mp["obs_desc"][o]]).sum())
                if d < best_d:
                    best_d, best_k = d, k
        if best_d < prop_d[best_k]:
            prop_d[best_k], prop_p[best_k] = best_d, p
    assert np.array_equal(out["prop_point"], prop_p)
    assert np.array_equal(out["prop_dist"], prop_d)
    assert np.array_equal(out["match_kp"], np.flatnonzero(prop_p >= 0))
    assert len(out["match_kp"]) > 20


# ------------------------------------------------------------- triangulation
def test_null_vector_vs_numpy_svd(oracle):
    rng = np.random.default_rng(4)
    for _ in range(200):
        A = rng.normal(size=(4, 4)) * 10 ** rng.uniform(-2, 3, size=(4, 1))
        v, s = oracle.null_vector4(A)
        U, S, Vt = np.linalg.svd(A)
        assert np.allclose(s, S, rtol=1e-12, atol=1e-13 * S[0])
        ref = Vt[3] * np.sign(Vt[3] @ v)
        gap = (S[2] - S[3]) / S[0]
        assert np.abs(v - ref).max() < 1e-13 / max(gap, 1e-6)


def test_triangulate_recovers_points_and_gates(oracle, synth):
    pr = synth.make_pair(1)
    a1, a2, ns = pr["truth12"]
    # true correspondences from the generator's permutations
    uv1, uv2 = pr["kp1"][a1[:ns]], pr["kp2"][a2[:ns]]
    out = oracle.triangulate(uv1, uv2, pr["poses"], pr["K"], min_parallax_cosine=1.0, max_reproj=4.0)
    # independent DLT via numpy SVD in float64
    K = pr["K"].astype(np.float64)
    Kmat = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]])
    P = [Kmat @ pr["poses"][i].reshape(4, 4)[:3].astype(np.float64) for i in range(2)]
    X = np.zeros((ns, 3))
    for i in range(ns):
        A = np.stack([uv1[i, 0] * P[0][2] - P[0][0], uv1[i, 1] * P[0][2] - P[0][1],
                      uv2[i, 0] * P[1][2] - P[1][0], uv2[i, 1] * P[1][2] - P[1][1]])
        v = np.linalg.svd(A)[2][3]
        X[i] = v[:3] / v[3]
    assert np.allclose(out["xyz"], X, rtol=2e-4, atol=1e-4)      # f32 P matrices / f32 output
    assert out["keep"].mean() > 0.9
    # default gates reject low-parallax points (cos > 0.9999)
    strict = oracle.triangulate(uv1, uv2, pr["poses"], pr["K"])
    assert strict["keep"].sum() < out["keep"].sum()
    assert np.array_equal(strict["out_index"], np.flatnonzero(strict["keep"]))
    # a point behind the cameras is dropped: swap the views' pixels
    bad = oracle.triangulate(uv2[:20], uv1[:20], pr["poses"], pr["K"], min_parallax_cosine=1.0, max_reproj=1e9)
    assert bad["keep"].sum() < 20
    assert len(oracle.triangulate(uv1[:0], uv2[:0], pr["poses"], pr["K"])["out_index"]) == 0


# ------------------------------------------------------------------ rotation
def test_triangulate_tracks_vs_numpy(oracle, synth):
    """Body of Mapper::triangulate_tracks (src/Mapper.cpp:246-305) against an independent float64 numpy
    formulation: same statuses away from the thresholds, same parallax terms, and the selection rule
    (threshold, quota top-up by ascending cosine, ties in track order) recomputed from the per-track values."""
    for kw, quota in ((dict(n_tracks=600, config_id=6), 100), (dict(n_tracks=260, far_frac=0.9, config_id=7), 100)):
        sc = synth.make_tracks(**kw)
        d = oracle.triangulate_tracks(sc["track_uv"], sc["sight_ptr"], sc["sight_pose"], sc["sight_uv"], sc["poses"],
                                      sc["kf_pose"], sc["K"], skip=sc["skip"], min_new_points=quota)
        fx, fy, cx, cy = [float(v) for v in sc["K"]]
        P = sc["poses"].reshape(-1, 4, 4).astype(np.float64)
        Tk = P[sc["kf_pose"]]
        ck = -Tk[:3, :3].T @ Tk[:3, 3]
        n = len(sc["track_uv"])
        first = sc["sight_pose"][np.minimum(sc["sight_ptr"][:-1], len(sc["sight_pose"]) - 1)]
        tri = oracle.triangulate(sc["sight_uv"][np.minimum(sc["sight_ptr"][:-1], len(sc["sight_uv"]) - 1)], sc["track_uv"],
                                 sc["poses"], sc["K"], idx1=first, idx2=np.full(n, sc["kf_pose"]),
                                 min_parallax_cosine=1.0, max_reproj=4.0)
        n_checked = 0
        for t in range(n):
            s0, s1 = sc["sight_ptr"][t], sc["sight_ptr"][t + 1]
            if s1 <= s0 or sc["skip"][t] or not tri["keep"][t]:
                assert d["status"][t] == 0
                continue
            X = d["xyz"][t].astype(np.float64)
            errs = []
            for s in range(s0, s1):
                T = P[sc["sight_pose"][s]]
                pc = T[:3, :3] @ X + T[:3, 3]
                uv = np.array([fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy]) if pc[2] >= 0 else np.array([-1.0, -1.0])
                errs.append(np.linalg.norm(uv - sc["sight_uv"][s]))
            errs = np.array(errs)
            if np.any(np.abs(errs - 4.0) < 1e-3):
                continue                                     # too close to the 4 px threshold for a float64 referee
            assert d["status"][t] == (2 if np.any(errs > 4.0) else 1)
            if d["status"][t] != 1:
                continue
            Tf = P[sc["sight_pose"][s0]]
            cf = -Tf[:3, :3].T @ Tf[:3, 3]
            a, b = cf - X, ck - X
            cosv = a @ b / np.linalg.norm(a) / np.linalg.norm(b)
            turned = np.arccos(np.clip((np.trace(Tk[:3, :3] @ Tf[:3, :3].T) - 1.0) / 2.0, -1.0, 1.0))
            assert abs(d["parallax_cos"][t] - cosv) < 2e-6
            # (trace - 1) / 2 is evaluated in float in the reference: acos amplifies its rounding near 1
            assert abs(d["required_cos"][t] - min(np.float32(0.999848), np.cos(0.2 * turned))) < 2e-6
            n_checked += 1
        assert n_checked > 100
        cand = np.flatnonzero(d["status"] == 1)
        ok = d["parallax_cos"][cand] <= d["required_cos"][cand]
        acc, rej = cand[ok], cand[~ok]
        top = 0
        if len(acc) < quota and len(rej):
            order = np.lexsort((rej, d["parallax_cos"][rej]))
            top = min(quota - len(acc), len(rej))
            acc = np.concatenate([acc, rej[order[:top]]])
        assert np.array_equal(d["accepted"], acc) and d["n_topped_up"] == top
        assert np.array_equal(d["inconsistent"], np.flatnonzero(d["status"] == 2))
    assert top > 0        # the second scene exercises the quota
    # empty input and all-skipped input
    e = oracle.triangulate_tracks(np.zeros((0, 2), np.float32), np.zeros(1, np.int32), np.zeros(0, np.int32),
                                  np.zeros((0, 2), np.float32), sc["poses"], 0, sc["K"])
    assert len(e["accepted"]) == 0 and len(e["inconsistent"]) == 0
    z = oracle.triangulate_tracks(sc["track_uv"], sc["sight_ptr"], sc["sight_pose"], sc["sight_uv"], sc["poses"],
                                  sc["kf_pose"], sc["K"], skip=np.ones(len(sc["track_uv"]), np.uint8))
    assert not z["status"].any() and len(z["accepted"]) == 0


def _cull_scene(synth, n_kf=8, n_points=600, seed=3):
    """A BA window re-expressed as what Mapper::cull_points sees: f32 poses, f32 positions, observation CSR."""
    w = synth.make_ba_window(n_kf=n_kf, n_points=n_points, run_max=min(6, n_kf))
    poses = np.stack([synth.make_pose(synth.rodrigues(w["cams"][c, :3]).T, w["cams"][c, 3:]) for c in range(n_kf)])
    rng = np.random.default_rng(seed)
    pos = w["points"].astype(np.float32)
    bad = rng.random(n_points) < 0.1
    pos[bad] += rng.normal(0, 0.4, (int(bad.sum()), 3)).astype(np.float32)      # some points worth culling
    return dict(positions=pos, obs_ptr=w["obs_ptr"], obs_pose=w["obs_cam"], obs_uv=w["obs_uv"],
                poses=poses.reshape(n_kf, 16).astype(np.float32), K=w["K"])


def test_point_errors_vs_numpy(oracle, synth):
    sc = _cull_scene(synth)
    d = oracle.point_errors(sc["positions"], sc["obs_ptr"], sc["obs_pose"], sc["obs_uv"], sc["poses"], sc["K"])
    fx, fy, cx, cy = [float(v) for v in sc["K"]]
    P = sc["poses"].reshape(-1, 4, 4).astype(np.float64)
    means = np.zeros(len(sc["positions"]))
    tot = 0.0
    for p in range(len(means)):
        es = []
        for o in range(sc["obs_ptr"][p], sc["obs_ptr"][p + 1]):
            T = P[sc["obs_pose"][o]]
            pc = T[:3, :3] @ sc["positions"][p].astype(np.float64) + T[:3, 3]
            uv = np.array([fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy]) if pc[2] >= 0 else np.array([-1.0, -1.0])
            es.append(np.linalg.norm(uv - sc["obs_uv"][o]))
        means[p] = np.mean(es) if es else 0.0
        tot += np.sum(es)
    assert np.allclose(d["mean_err"], means, rtol=2e-5, atol=2e-4)
    safe = np.abs(means - 3.0) > 1e-3
    assert np.array_equal(d["cull"][safe], (means > 3.0)[safe].astype(np.uint8))
    assert np.array_equal(d["cull_idx"], np.flatnonzero(d["cull"]))
    assert 0 < d["cull"].sum() < len(means)
    assert d["n_obs"] == sc["obs_ptr"][-1] and d["err_sum"] == pytest.approx(tot, rel=1e-5)


def test_pack_unpack_pose_roundtrip(oracle, rs, synth):
    rng = np.random.default_rng(6)
    for i in range(200):
        aa = rng.normal(size=3)
        aa *= (1e-9 if i % 20 == 0 else rng.uniform(0.01, 2.5)) / np.linalg.norm(aa)   # away from pi: log is ill-conditioned there
        R = synth.rodrigues(aa)
        pose = synth.make_pose(R.T, rng.normal(size=3) * 5)
        cam = oracle.pack_pose(pose)
        assert np.allclose(cam[:3], synth.log_so3(pose[:3, :3].astype(np.float64)), atol=5e-6)
        assert np.allclose(cam[3:], -pose[:3, :3].T @ pose[:3, 3], atol=1e-5)
        back = oracle.unpack_pose(cam)
        assert np.allclose(back, pose, atol=5e-6)
        # the library's host implementation is the same arithmetic
        assert np.array_equal(rs.pack_pose(pose), cam)
        assert np.array_equal(rs.unpack_pose(cam), back)
    # rotation by pi: the trace < 0 branch of the quaternion conversion
    R = np.diag([1.0, -1.0, -1.0])
    pose = synth.make_pose(R.T, np.zeros(3))
    assert np.allclose(np.abs(oracle.pack_pose(pose)[:3]), [np.pi, 0, 0], atol=1e-6)


# ----------------------------------------------------------------------- BA
def _fd_jac(oracle, cam, pt, uv, K, h=1e-6):
    jc = np.zeros((2, 6)); jp = np.zeros((2, 3))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        jc[:, k] = (oracle.reprojection(cam + d, pt, uv, K)[0] - oracle.reprojection(cam - d, pt, uv, K)[0]) / (2 * h)
    for k in range(3):
        d = np.zeros(3); d[k] = h
        jp[:, k] = (oracle.reprojection(cam, pt + d, uv, K)[0] - oracle.reprojection(cam, pt - d, uv, K)[0]) / (2 * h)
    return jc, jp


def test_reprojection_jets_vs_finite_differences(oracle):
    rng = np.random.default_rng(7)
    K = np.array([1000, 1000, 960, 540], np.float32)
    for _ in range(100):
        cam = np.concatenate([rng.normal(size=3) * rng.uniform(0.01, 1.5), rng.normal(size=3)])
        pt = cam[3:] + rng.normal(size=3) + np.array([0, 0, 6.0])
        uv = rng.uniform(0, 1000, 2).astype(np.float32)
        r, jc, jp = oracle.reprojection(cam, pt, uv, K)
        fc, fp = _fd_jac(oracle, cam, pt, uv, K)
        assert np.allclose(jc, fc, rtol=1e-5, atol=1e-4)
        assert np.allclose(jp, fp, rtol=1e-5, atol=1e-4)


def _numpy_problem(w):
    """dense residual / jacobian of the whole problem through the oracle's per-observation functor"""
    obs_pt = np.repeat(np.arange(len(w["points"])), np.diff(w["obs_ptr"]))
    free = np.flatnonzero(w["cam_free"])
    col_of = {c: 6 * i for i, c in enumerate(free)}
    return obs_pt, free, col_of


def test_first_lm_step_vs_dense_numpy(oracle, synth):
    """One LM iteration: Schur-eliminated solve == dense damped normal equations in numpy,
    with Ceres' Jacobi scaling and diagonal clamping applied exactly as documented."""
    w = synth.make_ba_window(n_kf=4, n_points=40, run_max=4, config_id=11)
    o = oracle.default_options(); o.max_num_iterations = 1
    cams, pts, s = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"],
                                        w["obs_uv"], w["K"], o)
    obs_pt, free, col_of = _numpy_problem(w)
    M, nc, P = len(w["obs_cam"]), 6 * len(free), len(w["points"])
    J = np.zeros((2 * M, nc + 3 * P)); r = np.zeros(2 * M)
    a = 5.991 ** 0.5
    cost = 0.0
    for oi in range(M):
        c, p = w["obs_cam"][oi], obs_pt[oi]
        ri, jc, jp = oracle.reprojection(w["cams"][c], w["points"][p], w["obs_uv"][oi], w["K"])
        sq = ri @ ri
        rho1 = 1.0 if sq <= a * a else a / np.sqrt(sq)
        cost += 0.5 * (sq if sq <= a * a else 2 * a * np.sqrt(sq) - a * a)
        sr = np.sqrt(rho1)
        r[2 * oi:2 * oi + 2] = sr * ri
        if c in col_of:
            J[2 * oi:2 * oi + 2, col_of[c]:col_of[c] + 6] = sr * jc
        J[2 * oi:2 * oi + 2, nc + 3 * p:nc + 3 * p + 3] = sr * jp
    assert np.isclose(cost, s["initial_cost"], rtol=1e-12)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    Js = J * scale
    D2 = np.clip((Js * Js).sum(0), 1e-6, 1e32) / 1e4
    y = np.linalg.solve(Js.T @ Js + np.diag(D2), Js.T @ r)
    delta = -y * scale
    exp_c = w["cams"].copy()
    for c, col in col_of.items():
        exp_c[c] += delta[col:col + 6]
    exp_p = w["points"] + delta[nc:].reshape(-1, 3)
    assert s["successful_steps"] == 1
    assert np.allclose(cams, exp_c, rtol=1e-9, atol=1e-11)
    assert np.allclose(pts, exp_p, rtol=1e-9, atol=1e-10)


def test_ba_converges_to_scipy_minimum(oracle, synth):
    """Run to convergence: same minimiser as scipy.optimize.least_squares on the identical
    robust objective (Huber on the 2-vector residual norm, f_scale = sqrt(5.991))."""
    from scipy.optimize import least_squares
    w = synth.make_ba_window(n_kf=4, n_points=30, run_max=4, config_id=12, outlier_frac=0.05)
    o = oracle.default_options(); o.max_num_iterations = 200
    o.function_tolerance = 1e-15; o.parameter_tolerance = 1e-14
    cams, pts, s = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"],
                                        w["obs_uv"], w["K"], o)
    obs_pt, free, col_of = _numpy_problem(w)
    nc = 6 * len(free)

    def norms(x):
        c = w["cams"].copy()
        for cam, col in col_of.items():
            c[cam] = x[col:col + 6]
        p = x[nc:].reshape(-1, 3)
        return np.array([np.linalg.norm(oracle.reprojection(c[w["obs_cam"][i]], p[obs_pt[i]], w["obs_uv"][i], w["K"])[0])
                         for i in range(len(obs_pt))])
    x0 = np.concatenate([cams[free].ravel(), pts.ravel()])
    a = 5.991 ** 0.5
    cost_at = lambda nn: 0.5 * np.sum(np.where(nn <= a, nn ** 2, 2 * a * nn - a * a))
    assert np.isclose(cost_at(norms(x0)), s["final_cost"], rtol=1e-9)
    sol = least_squares(norms, x0, loss="huber", f_scale=a, xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=50)
    assert sol.cost <= s["final_cost"] * (1 + 1e-9)
    assert sol.cost >= s["final_cost"] * (1 - 1e-6)      # the oracle had already reached the minimum


def test_ba_accept_rule_and_fixed_blocks(oracle, synth):
    w = synth.make_ba_window(n_kf=5, n_points=60, run_max=4)
    cams, pts, s = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"],
                                        w["obs_uv"], w["K"])
    assert s["usable"] == 1 and s["final_cost"] <= s["initial_cost"] and s["iterations"] <= 10
    assert np.array_equal(cams[:2], w["cams"][:2])            # fixed frames keep their blocks
    bad = w["points"].copy(); bad[:] = np.nan
    c2, p2, s2 = oracle.bundle_adjust(w["cams"], w["cam_free"], bad, w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    assert s2["usable"] == 0 and np.array_equal(c2, w["cams"])   # rejected solve writes nothing (:136-141)


def test_refine_pose_vs_ba_with_constant_points(oracle, synth):
    w = synth.make_ba_window(n_kf=3, n_points=300, run_min=3, run_max=3)
    sel = np.flatnonzero(w["obs_cam"] == 2)
    obs_pt = np.repeat(np.arange(300), 3)[sel]
    pts = w["points_true"][obs_pt]
    cam, s = oracle.refine_pose(w["cams"][2], pts, w["obs_uv"][sel], w["K"])
    assert s["usable"] == 1 and s["final_cost"] < 0.2 * s["initial_cost"]
    assert np.abs(cam - w["cams_true"][2]).max() < 5e-3
    cam2, s2 = oracle.refine_pose(w["cams"][2], pts[:0], w["obs_uv"][:0], w["K"])
    assert s2["usable"] == 0 and np.array_equal(cam2, w["cams"][2])


# ------------------------------------------------------------- local window
def _window_case(n_kf, window, new_frame, rng):
    n_pts = 40
    frames = n_kf + 1
    obs = [sorted(rng.choice(n_kf, size=min(n_kf, rng.integers(1, 4)), replace=False).tolist()) for _ in range(n_pts)]
    frame_pts = [[] for _ in range(frames)]
    for p, ks in enumerate(obs):
        for k in ks:
            frame_pts[k].append(p)
    if new_frame < 0:
        frame_pts[n_kf] = sorted(rng.choice(n_pts, 10, replace=False).tolist())
    fptr = np.cumsum([0] + [len(x) for x in frame_pts]).astype(np.int32)
    fpt = np.array([p for x in frame_pts for p in x], np.int32)
    pptr = np.cumsum([0] + [len(x) for x in obs]).astype(np.int32)
    pobs = np.array([k for x in obs for k in x], np.int32)
    return fptr, fpt, pptr, pobs, frame_pts, obs


@pytest.mark.parametrize("n_kf,window,new_in,fix", [(3, 20, True, False), (25, 20, True, False),
                                                   (25, 20, False, True), (30, 5, True, True), (2, 20, False, False)])
def test_build_local_window(oracle, rs, n_kf, window, new_in, fix):
    rng = np.random.default_rng(n_kf * 7 + window)
    new_frame = n_kf - 1 if new_in else -1
    fptr, fpt, pptr, pobs, frame_pts, obs = _window_case(n_kf, window, new_frame, rng)
    of, oo = oracle.build_local_window(n_kf, new_frame, window, fix, fptr, fpt, pptr, pobs)
    # python set restatement of src/LocalWindow.cpp:10-52
    self_id = new_frame if new_frame >= 0 else n_kf
    first = n_kf - window if n_kf > window else 2
    win = {self_id} | set(range(first, n_kf))
    anchors = {k for f in win for p in frame_pts[f] for k in obs[p] if k not in win}
    exp = []
    inc = False
    for i in range(n_kf):
        fixed = i < 2 or (fix and i == first)
        if i in win:
            exp.append((i, int(not fixed))); inc = inc or i == self_id
        elif fixed or i in anchors:
            exp.append((i, 0))
    if not inc:
        exp.append((self_id, 1))
    assert list(zip(of.tolist(), oo.tolist())) == exp
    lf, lo = rs.build_local_window(n_kf, new_frame, window, fix, fptr, fpt, pptr, pobs)
    assert np.array_equal(lf, of) and np.array_equal(lo, oo)


def test_reanchor_points_vs_numpy(oracle, rs, synth):
    """The tail of Mapper::bundle_adjust (src/Mapper.cpp:380-393) against a float32 numpy restatement, and the
    batched pose helpers of the library against the single-pose ones."""
    rng = np.random.default_rng(5)
    w = synth.make_ba_window(n_kf=6, n_points=50, run_max=4)
    before = np.stack([rs.unpack_pose(c) for c in w["cams"]]).astype(np.float32)
    after = np.stack([rs.unpack_pose(c) for c in w["cams_true"]]).astype(np.float32)
    n = 300
    fi = rng.integers(0, 6, n).astype(np.int32)
    X = rng.normal(0, 5, (n, 3)).astype(np.float32)
    out = oracle.reanchor_points(None, fi, before.reshape(-1, 16), after.reshape(-1, 16), X)
    for i in range(n):
        B, A = before[fi[i]], after[fi[i]]
        c = (B[:3, :3].astype(np.float64) @ X[i].astype(np.float64) + B[:3, 3]).astype(np.float32)
        ref = A[:3, :3].T.astype(np.float64) @ (c - A[:3, 3]).astype(np.float64)
        assert np.allclose(out[i], ref, rtol=2e-6, atol=2e-6)
    # identity when nothing moved
    same = oracle.reanchor_points(None, fi, before.reshape(-1, 16), before.reshape(-1, 16), X)
    assert np.allclose(same, X, atol=5e-6)
    # index list form
    idx = rng.permutation(n)[:100].astype(np.int32)
    part = oracle.reanchor_points(idx, fi[idx], before.reshape(-1, 16), after.reshape(-1, 16), X)
    assert np.array_equal(part[idx], out[idx])
    rest = np.setdiff1d(np.arange(n), idx)
    assert np.array_equal(part[rest], X[rest])
    # rs_unpack_poses == rs_unpack_pose per frame, masked
    mask = (np.arange(6) % 2).astype(np.uint8)
    buf = np.zeros((6, 16), np.float32)
    rs.unpack_poses(w["cams"], mask, buf)
    for c in range(6):
        assert np.array_equal(buf[c], rs.unpack_pose(w["cams"][c]).reshape(16) if mask[c] else np.zeros(16, np.float32))


def _fd(f, x, h=1e-6):
    x = np.array(x, float)
    cols = []
    for k in range(len(x)):
        xp, xm = x.copy(), x.copy()
        xp[k] += h
        xm[k] -= h
        cols.append((f(xp) - f(xm)) / (2 * h))
    return np.array(cols).T


def test_inertial_factor_jets_vs_finite_differences(oracle, synth):
    """oracle/imu.c: the 24-wide jets through PreintegrationError (src/ImuFactor.cpp:27-81), BiasRandomWalk (:89-118)
    and PredictedRotationError (src/Optimization.cpp:75-94) against central differences; the whitener against its
    definition; the residual vanishing (up to the synthetic noise) on the ground-truth trajectory."""
    w = synth.make_ba_window(n_kf=8, n_points=100, run_max=5, config_id=61)
    imu = synth.make_imu(w)
    assert len(imu["cam_i"]) == 5
    for f in (0, 3):
        i, j = int(imu["cam_i"][f]), int(imu["cam_j"][f])
        x0 = np.concatenate([w["cams"][i], imu["cam_velocity"][i], imu["cam_bias"][i], w["cams"][j], imu["cam_velocity"][j]])
        fun = lambda x: oracle.imu_preintegration(imu, f, x[0:6], x[6:9], x[9:15], x[15:21], x[21:24])[0]   # noqa: E731
        r, J = oracle.imu_preintegration(imu, f, x0[0:6], x0[6:9], x0[9:15], x0[15:21], x0[21:24])
        assert np.abs(J - _fd(fun, x0)).max() < 1e-6 * np.abs(J).max()
        W = oracle.imu_whitener(imu["covariance"][f].reshape(9, 9))
        assert np.allclose(W @ imu["covariance"][f].reshape(9, 9) @ W.T, np.eye(9), atol=1e-9)
        assert np.allclose(np.triu(W, 1), 0)
        rt, _ = oracle.imu_preintegration(imu, f, w["cams_true"][i], imu["cam_velocity_true"][i],
                                          np.concatenate([imu["bias_gyro"][f], imu["bias_accel"][f]]), w["cams_true"][j],
                                          imu["cam_velocity_true"][j])
        assert np.abs(rt).max() < 8.0            # whitened noise: a few sigma
        bw, Jb = oracle.imu_bias_walk(imu, f, imu["cam_bias"][i], imu["cam_bias"][j])
        fb = lambda x: oracle.imu_bias_walk(imu, f, x[:6], x[6:])[0]   # noqa: E731
        assert np.allclose(Jb, _fd(fb, np.concatenate([imu["cam_bias"][i], imu["cam_bias"][j]]), h=1e-7), rtol=1e-5, atol=1e-3)
    # a covariance that is not positive definite -> identity whitener (src/ImuFactor.cpp:13-15)
    assert np.array_equal(oracle.imu_whitener(-np.eye(9)), np.eye(9))
    # rotation prior, incl. a large rotation (trace < 0 branch of RotationMatrixToQuaternion) and the zero rotation
    rng = np.random.default_rng(3)
    for aa in (rng.normal(0, 0.3, 3), np.array([2.9, 0.3, -0.4]), np.zeros(3), np.array([1e-9, 0, 0])):
        pred = synth.rodrigues(rng.normal(0, 0.2, 3))
        pose = np.concatenate([aa, rng.normal(0, 1, 3)])
        r, J = oracle.rotation_prior(pred, 0.02, pose)
        fr = lambda x: oracle.rotation_prior(pred, 0.02, x)[0]   # noqa: E731
        assert np.allclose(J, _fd(fr, pose, h=1e-7), rtol=2e-5, atol=2e-4)
        assert np.allclose(J[:, 3:], 0)
        R = synth.rodrigues(aa) if np.linalg.norm(aa) > 1e-7 else np.eye(3) + np.array([[0, -aa[2], aa[1]], [aa[2], 0, -aa[0]], [-aa[1], aa[0], 0]])
        assert np.allclose(synth.rodrigues(r * 0.02) if np.linalg.norm(r) > 1e-7 else np.eye(3), pred.T @ R, atol=1e-7)


def test_inertial_bundle_adjust_reduces_to_vision_only_without_factors(oracle, synth):
    """No factor pairs -> the inertial entry point IS bundle_adjust (InertialInput::usable() false, :317)."""
    w = synth.make_ba_window(n_kf=6, n_points=200, run_max=5)
    imu = synth.make_imu(w)
    empty = dict(imu)
    for k in ("cam_i", "cam_j", "duration", "rotation", "velocity", "position", "covariance", "bias_gyro", "bias_accel", "bias_jacobian"):
        empty[k] = imu[k][:0]
    args = (w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    c, p, v, b, s = oracle.bundle_adjust_inertial(*args, empty)
    c0, p0, s0 = oracle.bundle_adjust(*args)
    assert s == s0 and np.array_equal(c, c0) and np.array_equal(p, p0)
    assert np.array_equal(v, imu["cam_velocity"]) and np.array_equal(b, imu["cam_bias"])
    # with the factors the velocities move towards the truth
    c, p, v, b, s = oracle.bundle_adjust_inertial(*args, imu)
    fr = np.flatnonzero(w["cam_free"])
    assert s["usable"] == 1
    assert np.abs(v - imu["cam_velocity_true"])[fr].max() < 0.7 * np.abs(imu["cam_velocity"] - imu["cam_velocity_true"])[fr].max()


def test_parallax_requirements_host_function_equals_the_oracle_bit_for_bit(rs, oracle, synth):
    """rs_parallax_requirements (host code of the product, no GPU): the rotation-dependent parallax requirement per
    first-sighting pose (reference src/Mapper.cpp:281-288) through the host's libm — must equal, bit for bit, what the
    oracle computes per track, so that K6 with this table selects exactly the CPU path's tracks."""
    for kw in (dict(n_tracks=2000, config_id=6), dict(n_tracks=5000, n_frames=16, max_sightings=14, config_id=9)):
        sc = synth.make_tracks(**kw)
        ref = oracle.triangulate_tracks(sc["track_uv"], sc["sight_ptr"], sc["sight_pose"], sc["sight_uv"], sc["poses"],
                                        sc["kf_pose"], sc["K"], skip=sc["skip"])
        req = rs.parallax_requirements(sc["poses"], sc["kf_pose"])
        cand = np.flatnonzero(ref["status"] == 1)
        assert len(cand) > 100
        first_pose = sc["sight_pose"][sc["sight_ptr"][cand]]
        assert np.array_equal(req[first_pose].view(np.uint32), ref["required_cos"][cand].view(np.uint32))
        assert np.all(req <= np.float32(0.999848))


def test_long_track_window_is_ill_conditioned(oracle, synth):
    """The window test_bundle_adjust_banded_reduced_solve accepts at an ABSOLUTE tolerance (5e-6 on cameras) instead of
    1e-7 relative: its reduced camera system is numerically singular undamped and has condition > 1e6 at the final trust-region
    radius, so two correct f64 solves may differ by 1e-7 .. 1e-6 (measured differences: tools/tol_check.py)."""
    import dense_lm
    w = synth.make_ba_window(n_kf=40, n_points=4000, run_min=3, run_max=24, config_id=143)
    rc, rp, s = oracle.bundle_adjust(w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    cond = dense_lm.reduced_system_condition(w, rc, rp, s["final_radius"])
    assert cond["final radius"][0] > 1e6 and cond["undamped"][1] < 1e-6 * cond["undamped"][2], cond
    # the benchmark window for comparison: well conditioned, held to 1e-7
    w3 = synth.make_ba_window(n_kf=8, n_points=300, run_max=5, config_id=3)
    rc3, rp3, s3 = oracle.bundle_adjust(w3["cams"], w3["cam_free"], w3["points"], w3["obs_ptr"], w3["obs_cam"], w3["obs_uv"], w3["K"])
    cond3 = dense_lm.reduced_system_condition(w3, rc3, rp3, s3["final_radius"])
    assert cond3["final radius"][0] < cond["final radius"][0]
