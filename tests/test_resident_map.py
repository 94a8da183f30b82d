"""§8(f) rank 4: the device-resident map (rs_map / rs_frame).  The resident path must give exactly what the flattened
path gives — rs_reproj_match on arrays rebuilt from scratch (itself checked against the oracle) — before and after
incremental updates that mirror Map::create_point / remove_point / associate / set_position / Frame::set_pose
(reference src/Map.cpp:44-124)."""
import numpy as np
import pytest

from conftest import to_np

pytestmark = pytest.mark.gpu


def centre_f32(T):
    """Frame::camera_center in f32 with the library's operation order: (-R0i*t0 + -R1i*t1) + -R2i*t2"""
    T = np.asarray(T, np.float32).reshape(16)
    return np.array([(-T[i] * T[3] + -T[4 + i] * T[7]) + -T[8 + i] * T[11] for i in range(3)], np.float32)


class Scene:
    """Host-side model of the reference's objects: key frames with their own descriptor matrices, points with ordered
    observation lists; builds the resident map incrementally and the flat arrays from scratch."""

    def __init__(self, ctx, rs, synth, n_kf=6, n_points=600, seed=3, dup_pairs=False):
        self.ctx, self.rs = ctx, rs
        w = synth.make_ba_window(n_kf=n_kf, n_points=n_points, run_max=5, config_id=70 + seed)
        frame, mp = synth.make_match_scene(w, n_keypoints=500, kdtree_build=rs.kdtree_build, config_id=70 + seed)
        self.w, self.frame, self.K = w, frame, w["K"]
        self.map = rs.ResidentMap(ctx)
        self.kf_pose, self.kf_desc, self.kf_kp, self.kf_handle = [], [], [], []
        obs_pt = np.repeat(np.arange(n_points), np.diff(w["obs_ptr"]))
        pool_of_obs = mp["desc_pool"][mp["obs_desc"]]                 # descriptor of every observation, CSR order
        if dup_pairs:      # points 2k and 2k + 1 carry the SAME descriptors: they tie for every keypoint they both reach
            for p in range(0, n_points - 1, 2):
                a0, a1, b0, b1 = w["obs_ptr"][p], w["obs_ptr"][p + 1], w["obs_ptr"][p + 1], w["obs_ptr"][p + 2]
                for j in range(b0, b1):
                    pool_of_obs[j] = pool_of_obs[a0 + (j - b0) % (a1 - a0)]
                if a1 - a0 > b1 - b0:      # same descriptor SET on both sides: cut the longer one down to the shorter's rows
                    for j in range(a0, a1):
                        pool_of_obs[j] = pool_of_obs[a0 + (j - a0) % (b1 - b0)]
        kp_index = np.zeros(len(obs_pt), np.int64)
        for k in range(n_kf):
            sel = np.flatnonzero(w["obs_cam"] == k)
            kp_index[sel] = np.arange(len(sel))
            # a few unmatched extra keypoints so that key frames are not exactly their observations
            extra = 7
            rng = np.random.default_rng(100 + k)
            desc = np.concatenate([pool_of_obs[sel], rng.integers(0, 256, (extra, 32), dtype=np.uint8)])
            kp = np.concatenate([w["obs_uv"][sel], rng.uniform(0, 500, (extra, 2)).astype(np.float32)])
            pose = w["poses_true"][k].astype(np.float32)
            fr = rs.ResidentFrame(ctx, kp, desc)
            self.kf_handle.append(self.map.add_keyframe(fr, pose))
            fr.close()
            self.kf_pose.append(pose.reshape(16).copy()); self.kf_desc.append(desc); self.kf_kp.append(kp)
        self.pos, self.alive, self.obs = [], [], []
        for p in range(n_points):
            h = self.map.add_point(mp["positions"][p])
            assert h == p
            self.pos.append(mp["positions"][p].astype(np.float32)); self.alive.append(1); self.obs.append([])
            for o in range(w["obs_ptr"][p], w["obs_ptr"][p + 1]):
                self.associate(p, int(w["obs_cam"][o]), int(kp_index[o]))
        self.rframe = rs.ResidentFrame(ctx, frame["keypoints"], frame["descriptors"])

    # -- the update calls, applied to both the python model and the resident map
    def associate(self, p, kf, kp):
        self.obs[p] = [o for o in self.obs[p] if o[0] != kf]
        for q in range(len(self.obs)):
            if q != p:
                self.obs[q] = [o for o in self.obs[q] if not (o[0] == kf and o[1] == kp)]
        self.obs[p].append((kf, kp))
        self.map.add_observation(p, kf, kp)

    def add_point(self, xyz):
        h = self.map.add_point(xyz)
        self.pos.append(np.asarray(xyz, np.float32)); self.alive.append(1); self.obs.append([])
        return h

    def remove_point(self, p):
        self.alive[p] = 0
        self.obs[p] = []
        self.map.remove_point(p)

    def set_position(self, p, xyz):
        self.pos[p] = np.asarray(xyz, np.float32)
        self.map.set_position(p, xyz)

    def set_pose(self, kf, pose):
        self.kf_pose[kf] = np.asarray(pose, np.float32).reshape(16).copy()
        self.map.set_keyframe_pose(kf, pose)

    # -- flat arrays from scratch
    def flat(self, matched_points=(), required=-1, only=None):
        P = len(self.pos)
        rows = np.cumsum([0] + [len(d) for d in self.kf_desc])
        elig = np.array(self.alive, np.uint8)
        elig[list(matched_points)] = 0
        if only is not None:
            m = np.zeros(P, np.uint8); m[list(only)] = 1
            elig &= m
        optr, okf, odesc = [0], [], []
        for p in range(P):
            if required >= 0 and not any(o[0] == required for o in self.obs[p]):
                elig[p] = 0
            for kf, kp in self.obs[p]:
                okf.append(kf); odesc.append(rows[kf] + kp)
            optr.append(len(okf))
        return dict(positions=np.array(self.pos, np.float32).reshape(-1, 3), eligible=elig, obs_ptr=np.array(optr, np.int32),
                    obs_kf=np.array(okf, np.int32), obs_desc=np.array(odesc, np.int32),
                    kf_centers=np.stack([centre_f32(T) for T in self.kf_pose]), desc_pool=np.concatenate(self.kf_desc))

    def check(self, oracle, kp_matched=None, matched_points=(), required=-1, only=None, replace=0):
        fr = dict(self.frame)
        if kp_matched is not None:
            fr["kp_matched"] = kp_matched
        else:
            fr["kp_matched"] = np.zeros(len(fr["keypoints"]), np.uint8)
        mp = self.flat(matched_points, required, only)
        ref = oracle.reproj_match(fr, mp, replace=replace)
        fv, k1 = self.ctx.make_frame_view(fr)
        mv, k2 = self.ctx.make_map_view(mp)
        flat = self.ctx.reproj_match(fv, mv, replace=replace)
        n = int(to_np(flat["count"])[0])
        assert np.array_equal(to_np(flat["match_kp"])[:n], ref["match_kp"]) and np.array_equal(to_np(flat["match_point"])[:n], ref["match_point"])
        mk, mpt = self.map.match(self.rframe, fr["pose"], self.K, fr["width"], fr["height"], kp_matched=kp_matched,
                                 matched_points=matched_points, required_observer=required, only_points=only, replace=replace)
        assert np.array_equal(mk, ref["match_kp"]) and np.array_equal(mpt, ref["match_point"])
        return len(mk)


def test_resident_map_matches_equal_the_flattened_path(ctx, rs, oracle, synth):
    sc = Scene(ctx, rs, synth)
    c = sc.map.counts()
    assert c["alive"] == 600 and c["key_frames"] == 6 and c["observations"] == sum(len(o) for o in sc.obs)
    n_all = sc.check(oracle)                                             # match_map
    assert n_all > 50
    n_kf = sc.check(oracle, required=5)                                  # match_key_frame (last key frame)
    assert 0 < n_kf <= n_all
    # the frame already matches some keypoints / points (src/MapMatcher.cpp:53,81)
    rng = np.random.default_rng(1)
    kpm = (rng.random(len(sc.frame["keypoints"])) < 0.3).astype(np.uint8)
    pts = rng.choice(600, 80, replace=False)
    sc.check(oracle, kp_matched=kpm, matched_points=pts)
    sc.check(oracle, kp_matched=kpm, matched_points=pts, required=4)
    sc.check(oracle)                                                     # the flag table was left clean
    # match_for_fuse: explicit list, already-matched keypoints stay eligible
    only = np.sort(rng.choice(600, 200, replace=False))
    sc.check(oracle, kp_matched=kpm, only=only, replace=1)
    sc.map.close()


def test_resident_map_incremental_updates(ctx, rs, oracle, synth):
    sc = Scene(ctx, rs, synth, seed=5)
    rng = np.random.default_rng(2)
    before = sc.check(oracle)
    # cull points, move points (BA write-back), move a key frame
    for p in rng.choice(600, 60, replace=False):
        sc.remove_point(int(p))
    for p in np.flatnonzero(sc.alive)[:100]:
        sc.set_position(int(p), sc.pos[p] + rng.normal(0, 0.01, 3).astype(np.float32))
    T = np.array(sc.kf_pose[3]).reshape(4, 4).copy()
    T[:3, 3] += np.array([0.02, -0.01, 0.03], np.float32)
    sc.set_pose(3, T)
    after = sc.check(oracle)
    assert after <= before
    # new points seen by the last two key frames, re-association of an occupied keypoint (Map::associate semantics)
    for i in range(40):
        src = int(np.flatnonzero(sc.alive)[i])
        h = sc.add_point(sc.pos[src] + np.float32(0.05))
        sc.associate(h, 5, i)                        # keypoint i of key frame 5 is taken from whoever had it
        sc.associate(h, 4, len(sc.kf_kp[4]) - 1 - (i % 7))
    sc.check(oracle)
    sc.check(oracle, required=5)
    # capacity growth: enough new points to outgrow the device arrays
    for i in range(5000):
        h = sc.add_point(rng.normal(0, 30, 3).astype(np.float32))
    sc.associate(h, 0, 0)
    sc.check(oracle)
    assert sc.map.counts()["alive"] == sum(sc.alive)
    sc.map.close()


def test_resident_map_bundle_adjust_equals_flat_solve(ctx, rs, oracle, synth):
    """rs_map_bundle_adjust flattens the window on the device from the resident image; the flat problem built here from the python
    model is the same, so poses and points must agree to the last bits of the solver's own noise, and the map must
    carry the result afterwards (checked through a match against the flat path with the new state)."""
    sc = Scene(ctx, rs, synth, n_kf=7, n_points=500, seed=7)
    w = sc.w
    # perturb what the map holds: BA has something to do
    rng = np.random.default_rng(4)
    for k in range(2, 7):
        T = np.array(sc.kf_pose[k]).reshape(4, 4).copy()
        T[:3, 3] += rng.normal(0, 0.01, 3).astype(np.float32)
        sc.set_pose(k, T)
    kfs = np.arange(7, dtype=np.int32)
    free = np.array([0, 0, 1, 1, 1, 1, 1], np.uint8)
    # flat problem from the python model (frame order, keypoint order: Frame::map_matches ascending keypoint index)
    kp_point = [dict() for _ in range(7)]
    for p, ol in enumerate(sc.obs):
        for kf, kp in ol:
            kp_point[kf][kp] = p
    order, pid = [], {}
    for c in range(7):
        if free[c]:
            for kp in sorted(kp_point[c]):
                p = kp_point[c][kp]
                if p not in pid and len(sc.obs[p]) >= 2:
                    pid[p] = len(order); order.append(p)
    # the library lists the free points in ascending slot order (the reference walks its frames' match tables: same set)
    order = sorted(order)
    pid = {p: i for i, p in enumerate(order)}
    per = [[] for _ in order]
    for c in range(7):
        for kp in sorted(kp_point[c]):
            p = kp_point[c][kp]
            if p in pid:
                per[pid[p]].append((c, sc.kf_kp[c][kp]))
    obs_ptr = np.cumsum([0] + [len(x) for x in per]).astype(np.int32)
    obs_cam = np.array([c for x in per for c, _ in x], np.int32)
    obs_uv = np.array([uv for x in per for _, uv in x], np.float32)
    cams = np.stack([rs.pack_pose(np.array(T).reshape(4, 4)) for T in sc.kf_pose])
    pts = np.array([sc.pos[p] for p in order], np.float64)
    dc, dp = ctx.dev(cams), ctx.dev(pts)
    s_flat = ctx.bundle_adjust(dc, free, dp, ctx.dev(obs_ptr), ctx.dev(obs_cam), ctx.dev(obs_uv), sc.K)
    s, poses, out_pts, out_xyz = sc.map.bundle_adjust(kfs, free, sc.K)
    assert s["usable"] == 1 and (s["iterations"], s["successful_steps"]) == (s_flat["iterations"], s_flat["successful_steps"])
    assert np.isclose(s["final_cost"], s_flat["final_cost"], rtol=1e-9)
    assert np.array_equal(out_pts, np.array(order, np.int32))
    assert np.allclose(out_xyz, to_np(dp).astype(np.float32), rtol=1e-6, atol=1e-6)
    fc = to_np(dc)
    for c in range(7):
        ref = rs.unpack_pose(fc[c]).reshape(16) if free[c] else np.array(sc.kf_pose[c])
        assert np.allclose(poses[c], ref, rtol=1e-6, atol=1e-6)
    # the oracle agrees with the flat solve
    rc, rp, rs_ = oracle.bundle_adjust(cams, free, pts, obs_ptr, obs_cam, obs_uv, sc.K)
    assert (rs_["iterations"], rs_["successful_steps"]) == (s["iterations"], s["successful_steps"])
    assert np.allclose(fc, rc, rtol=1e-7, atol=1e-9)
    # the resident map now holds the result: bring the python model to the same state and compare a match
    for c in range(7):
        sc.kf_pose[c] = poses[c].copy()
    for p, x in zip(out_pts, out_xyz):
        sc.pos[int(p)] = x.copy()
    sc.check(oracle)
    sc.map.close()
