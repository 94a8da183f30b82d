"""Synthetic inputs for the hot path, shaped as SURVEY.md §8(d) / BASELINE.json configs.

All inputs are generated on the host with numpy's PCG64 (seed 0x5EED0000 +
config id) and handed unchanged to the GPU library, the oracle and the CPU
baseline, so every leg sees bit-identical arrays.  There is no dataset: the
reference's inputs are video frames; here keypoints are projections of random
landmarks and descriptors are random 256-bit strings with 5 % bit flips per
observation (true-match Hamming ~13, impostor ~128).
"""
import numpy as np

SEED_BASE = 0x5EED0000


def rng_for(config_id, stream=0):
    return np.random.default_rng([SEED_BASE + int(config_id), int(stream)])


def yaw_matrix(deg):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def make_pose(R_wc, centre):
    """world->camera 4x4 (f32) from camera-to-world rotation and centre."""
    T = np.eye(4)
    T[:3, :3] = R_wc.T
    T[:3, 3] = -R_wc.T @ centre
    return T.astype(np.float32)


def rodrigues(aa):
    th = np.linalg.norm(aa)
    if th < 1e-12:
        return np.eye(3)
    k = aa / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def log_so3(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-12:
        return np.zeros(3)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return w * th


def random_descriptors(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def flip_bits(rng, desc, p=0.05):
    bits = np.unpackbits(desc, axis=1)
    flips = rng.random(bits.shape) < p
    return np.packbits(bits ^ flips.astype(np.uint8), axis=1)


def project(T, K, X):
    Xc = X @ T[:3, :3].T.astype(np.float64) + T[:3, 3].astype(np.float64)
    u = K[0] * Xc[:, 0] / Xc[:, 2] + K[2]
    v = K[1] * Xc[:, 1] / Xc[:, 2] + K[3]
    return np.stack([u, v], 1), Xc[:, 2]


def pair_config(config_id):
    if config_id == 1:
        return dict(width=640, height=480, K=np.array([500, 500, 320, 240], np.float32), n=500, unmatched=0.0)
    return dict(width=1920, height=1080, K=np.array([1000, 1000, 960, 540], np.float32), n=2000, unmatched=0.10)


def make_pair(config_id=2, seed_stream=0, n=None, pose_jitter=0.0):
    """Two-frame scene of cfg 1 / cfg 2: returns a dict with
    desc1/kp1 (train, the keyframe), desc2/kp2 (query, the new frame), poses [2][16], K.
    pose_jitter > 0 (cfg 4 batches): the second pose varies with the seed stream."""
    cfg = pair_config(config_id)
    rng = rng_for(config_id, seed_stream)
    N = cfg["n"] if n is None else int(n)
    K, W, H = cfg["K"].astype(np.float64), cfg["width"], cfg["height"]
    T1 = make_pose(np.eye(3), np.zeros(3))
    if pose_jitter > 0.0:
        jr = np.random.default_rng(0xC4 + seed_stream)
        T2 = make_pose(yaw_matrix(2.0 + pose_jitter * jr.uniform(-1, 1)),
                       np.array([0.2, 0.0, 0.05]) + pose_jitter * 0.05 * jr.uniform(-1, 1, 3))
    else:
        T2 = make_pose(yaw_matrix(2.0), np.array([0.2, 0.0, 0.05]))
    n_shared = int(round(N * (1.0 - cfg["unmatched"])))
    # landmarks uniform in frame 1's frustum, depth U[4,20]
    z = rng.uniform(4, 20, n_shared)
    u = rng.uniform(0, W, n_shared)
    v = rng.uniform(0, H, n_shared)
    X = np.stack([(u - K[2]) / K[0] * z, (v - K[3]) / K[1] * z, z], 1)
    base = random_descriptors(rng, n_shared)
    kp1, _ = project(T1, K, X)
    kp2, _ = project(T2, K, X)
    kp1 = kp1 + rng.normal(0, 0.5, kp1.shape)
    kp2 = kp2 + rng.normal(0, 0.5, kp2.shape)
    d1 = flip_bits(rng, base)
    d2 = flip_bits(rng, base)
    n_extra = N - n_shared
    if n_extra > 0:
        for kp, d in ((kp1, d1), (kp2, d2)):
            pass
        e1 = np.stack([rng.uniform(0, W, n_extra), rng.uniform(0, H, n_extra)], 1)
        e2 = np.stack([rng.uniform(0, W, n_extra), rng.uniform(0, H, n_extra)], 1)
        kp1 = np.concatenate([kp1, e1]); kp2 = np.concatenate([kp2, e2])
        d1 = np.concatenate([d1, random_descriptors(rng, n_extra)])
        d2 = np.concatenate([d2, random_descriptors(rng, n_extra)])
    # independent shuffles so that index i in frame 1 is not index i in frame 2
    p1 = rng.permutation(N)
    p2 = rng.permutation(N)
    return dict(desc1=np.ascontiguousarray(d1[p1]), kp1=kp1[p1].astype(np.float32),
                desc2=np.ascontiguousarray(d2[p2]), kp2=kp2[p2].astype(np.float32),
                poses=np.stack([T1.reshape(16), T2.reshape(16)]).astype(np.float32),
                K=cfg["K"], width=W, height=H,
                truth12=(np.argsort(p1), np.argsort(p2), n_shared))


def make_pair_batch(batch=64, config_id=2, first_stream=0, n=None, pose_jitter=1.0):
    """cfg 4 (BASELINE.json configs[3]): `batch` independent instances of the cfg-2 pair, stacked:
    desc1/desc2 [B][N][32], kp1/kp2 [B][N][2], poses [B][2][16], K."""
    prs = [make_pair(config_id, seed_stream=first_stream + b, n=n, pose_jitter=pose_jitter) for b in range(batch)]
    out = {k: np.ascontiguousarray(np.stack([p[k] for p in prs])) for k in ("desc1", "kp1", "desc2", "kp2", "poses")}
    out.update(K=prs[0]["K"], width=prs[0]["width"], height=prs[0]["height"], pairs=prs)
    return out


def make_ba_window(n_kf=20, n_points=10000, config_id=3, seed_stream=0, n_fixed=2,
                   run_min=2, run_max=10, pixel_noise=0.5, outlier_frac=0.02,
                   rot_noise_deg=0.5, trans_noise=0.01, depth_noise=0.01):
    """cfg 3 / cfg 5 window: keyframes on a gently curving forward track, each landmark
    seen by a run of consecutive keyframes.  Returns the flat BA problem
    (cams [C][6] = angle-axis(R_cw) + centre, points [P][3], CSR observations) plus
    the ground truth and the keyframe poses."""
    rng = rng_for(config_id, seed_stream)
    K = np.array([1000, 1000, 960, 540], np.float32)
    Kd = K.astype(np.float64)
    W, H = 1920, 1080
    R = np.eye(3)
    c = np.zeros(3)
    Rs, cs = [], []
    for i in range(n_kf):
        Rs.append(R.copy()); cs.append(c.copy())
        R = R @ yaw_matrix(1.5)
        c = c + R @ np.array([0, 0, 0.5])
    poses_true = np.stack([make_pose(Rs[i], cs[i]) for i in range(n_kf)])
    run_max = min(run_max, n_kf)
    run_len = rng.integers(run_min, run_max + 1, n_points)
    start = (rng.random(n_points) * (n_kf - run_len + 1)).astype(np.int64)
    mid = start + run_len // 2
    z = rng.uniform(4, 20, n_points)
    u = rng.uniform(0.1 * W, 0.9 * W, n_points)
    v = rng.uniform(0.1 * H, 0.9 * H, n_points)
    Xc = np.stack([(u - Kd[2]) / Kd[0] * z, (v - Kd[3]) / Kd[1] * z, z], 1)
    Rm = np.stack([Rs[m] for m in mid]); cm = np.stack([cs[m] for m in mid])
    X = np.einsum("nij,nj->ni", Rm, Xc) + cm
    obs_ptr = np.zeros(n_points + 1, np.int32)
    obs_ptr[1:] = np.cumsum(run_len)
    M = int(obs_ptr[-1])
    obs_cam = np.concatenate([np.arange(s, s + l) for s, l in zip(start, run_len)]).astype(np.int32)
    obs_pt = np.repeat(np.arange(n_points), run_len)
    uv = np.zeros((M, 2))
    for k in range(n_kf):
        sel = obs_cam == k
        if sel.any():
            uv[sel], _ = project(poses_true[k], Kd, X[obs_pt[sel]])
    uv += rng.normal(0, pixel_noise, uv.shape)
    out = rng.random(M) < outlier_frac
    uv[out] += rng.uniform(-30, 30, (int(out.sum()), 2))
    # perturbed initial state
    cams = np.zeros((n_kf, 6))
    cams_true = np.zeros((n_kf, 6))
    for i in range(n_kf):
        Rcw = Rs[i].T
        cams_true[i, :3] = log_so3(Rcw); cams_true[i, 3:] = cs[i]
        if i < n_fixed:
            cams[i] = cams_true[i]
        else:
            dR = rodrigues(rng.normal(0, np.deg2rad(rot_noise_deg) / np.sqrt(3), 3))
            cams[i, :3] = log_so3(dR @ Rcw)
            cams[i, 3:] = cs[i] + rng.normal(0, trans_noise * 0.5, 3)
    cams = cams.astype(np.float32).astype(np.float64)   # pack_pose output is f32-valued
    cams_true = cams_true.astype(np.float32).astype(np.float64)
    depth_scale = 1.0 + rng.normal(0, depth_noise, n_points)
    pts = cm + (X - cm) * depth_scale[:, None]
    pts = pts.astype(np.float32).astype(np.float64)     # MapPoint::position is f32
    cam_free = np.ones(n_kf, np.uint8)
    cam_free[:n_fixed] = 0
    return dict(cams=cams, cam_free=cam_free, points=pts, obs_ptr=obs_ptr, obs_cam=obs_cam,
                obs_uv=uv.astype(np.float32), K=K, cams_true=cams_true, points_true=X,
                poses_true=poses_true, width=W, height=H, run_start=start.astype(np.int32),
                run_len=run_len.astype(np.int32))


def make_imu(window, seed=11, duration=0.5, sigma_rot=2e-3, sigma_vel=2e-2, sigma_pos=1e-2, skip=()):
    """Synthetic IMU factor pairs for a BA window (reference src/Optimization.cpp:317-346): one per pair of consecutive
    FREE cameras (pairs listed in `skip` are left out, like a gap with fewer than two samples), consistent with the
    ground-truth trajectory up to noise:
        delta R = R_i R_j^T,  delta v = R_i (v_j - v_i - g T),  delta p = R_i (c_j - c_i - v_i T - g T^2 / 2)
    (R = world -> camera; the residual of src/ImuFactor.cpp:66-69 vanishes on them).  Covariances are random SPD
    matrices of realistic scale, the bias Jacobians random, and the current bias estimates differ slightly from the
    biases used at preintegration so that the first-order bias correction (:49-61) is exercised.  Also returns the
    perturbed initial velocities / biases per camera (cam_velocity [C][3], cam_bias [C][6]) and gravity."""
    rng = np.random.default_rng(0x1A2B0000 + seed)
    cams_true = window["cams_true"]
    C = len(cams_true)
    T = float(duration)
    g = np.array([0.0, 0.0, -9.80665])
    Rcw = np.stack([rodrigues(cams_true[c, :3]) for c in range(C)])
    ctr = cams_true[:, 3:]
    v_true = np.zeros((C, 3))
    for c in range(C):
        n = min(c + 1, C - 1)
        p = max(n - 1, 0)
        v_true[c] = (ctr[n] - ctr[p]) / T if n != p else 0.0
    free = np.flatnonzero(window["cam_free"])
    fac = dict(cam_i=[], cam_j=[], duration=[], rotation=[], velocity=[], position=[], covariance=[], bias_gyro=[],
               bias_accel=[], bias_jacobian=[])
    for a, b in zip(free[:-1], free[1:]):
        if b != a + 1 or (int(a), int(b)) in skip:
            continue
        dR = Rcw[a] @ Rcw[b].T @ rodrigues(rng.normal(0, sigma_rot, 3))
        dv = Rcw[a] @ (v_true[b] - v_true[a] - g * T) + rng.normal(0, sigma_vel, 3)
        dp = Rcw[a] @ (ctr[b] - ctr[a] - v_true[a] * T - 0.5 * g * T * T) + rng.normal(0, sigma_pos, 3)
        A = rng.normal(0, 1, (9, 9)) * np.array([sigma_rot] * 3 + [sigma_vel] * 3 + [sigma_pos] * 3)[:, None] * 0.3
        cov = A @ A.T + np.diag(np.array([sigma_rot] * 3 + [sigma_vel] * 3 + [sigma_pos] * 3) ** 2)
        bj = rng.normal(0, 1, (9, 6)) * np.array([T] * 3 + [T] * 3 + [T * T] * 3)[:, None] * 0.5
        fac["cam_i"].append(a); fac["cam_j"].append(b); fac["duration"].append(T)
        fac["rotation"].append(dR.reshape(9)); fac["velocity"].append(dv); fac["position"].append(dp)
        fac["covariance"].append(cov.reshape(81)); fac["bias_gyro"].append(rng.normal(0, 1e-3, 3))
        fac["bias_accel"].append(rng.normal(0, 1e-2, 3)); fac["bias_jacobian"].append(bj.reshape(54))
    out = {k: np.array(v, np.int32 if k.startswith("cam_") else np.float64) for k, v in fac.items()}
    nF = len(out["cam_i"])
    bias = np.zeros((C, 6))
    for f in range(nF):
        i = out["cam_i"][f]
        bias[i, :3] = out["bias_gyro"][f] + rng.normal(0, 2e-4, 3)
        bias[i, 3:] = out["bias_accel"][f] + rng.normal(0, 2e-3, 3)
    if nF:
        bias[out["cam_j"][-1]] = bias[out["cam_i"][-1]] + rng.normal(0, 1e-4, 6)
    out.update(cam_velocity=v_true + rng.normal(0, 0.05, (C, 3)), cam_bias=bias, gravity=g, cam_velocity_true=v_true,
               gyro_bias_sigma=2.78e-5, accel_bias_sigma=2.79e-3)      # imu::NoiseDensity defaults, src/Imu.h:42-49
    return out


def shard_ba_by_landmark(prob, n_shards, shard, bounds=None):
    """Landmark shard `shard` of `n_shards` (contiguous blocks of landmarks;
    cameras replicated) — SURVEY.md §8(e).  bounds: explicit split points [n_shards + 1]
    (equal bounds give an empty shard)."""
    P = len(prob["points"])
    lo = (P * shard) // n_shards if bounds is None else int(bounds[shard])
    hi = (P * (shard + 1)) // n_shards if bounds is None else int(bounds[shard + 1])
    o0, o1 = int(prob["obs_ptr"][lo]), int(prob["obs_ptr"][hi])
    out = dict(prob)
    out["points"] = prob["points"][lo:hi].copy()
    out["obs_ptr"] = (prob["obs_ptr"][lo:hi + 1] - o0).astype(np.int32)
    out["obs_cam"] = prob["obs_cam"][o0:o1].copy()
    out["obs_uv"] = prob["obs_uv"][o0:o1].copy()
    out["point_range"] = (lo, hi)
    return out


def make_match_scene(window=None, n_keypoints=2000, config_id=3, seed_stream=7, matched_frac=0.3,
                     kdtree_build=None):
    """Reprojection-gated matching scene (a2): the newest frame of a BA window
    against the window's landmarks.  Every landmark observation gets a
    descriptor row in the pool; the frame sees a subset of the landmarks plus
    clutter keypoints; `matched_frac` of its keypoints are already matched."""
    if window is None:
        window = make_ba_window(config_id=config_id)
    rng = rng_for(config_id, seed_stream)
    K = window["K"].astype(np.float64)
    W, H = window["width"], window["height"]
    P = len(window["points"])
    n_kf = len(window["cams"])
    X = window["points_true"]
    # the frame: one step beyond the last keyframe
    T_last = window["poses_true"][-1].astype(np.float64)
    R_wc = T_last[:3, :3].T @ yaw_matrix(1.5)
    centre = -T_last[:3, :3].T @ T_last[:3, 3] + R_wc @ np.array([0, 0, 0.5])
    pose = make_pose(R_wc, centre)
    uv, zc = project(pose, K, X)
    vis = (zc > 0.5) & (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)
    vis_idx = np.flatnonzero(vis)
    n_seen = min(len(vis_idx), int(n_keypoints * 0.8))
    seen = rng.choice(vis_idx, n_seen, replace=False) if n_seen > 0 else np.zeros(0, np.int64)
    base = random_descriptors(rng, P)
    kp = uv[seen] + rng.normal(0, 1.5, (n_seen, 2))
    desc = flip_bits(rng, base[seen])
    n_extra = n_keypoints - n_seen
    kp = np.concatenate([kp, np.stack([rng.uniform(0, W, n_extra), rng.uniform(0, H, n_extra)], 1)])
    desc = np.concatenate([desc, random_descriptors(rng, n_extra)])
    perm = rng.permutation(n_keypoints)
    kp = kp[perm].astype(np.float32)
    desc = np.ascontiguousarray(desc[perm])
    kp_matched = (rng.random(n_keypoints) < matched_frac).astype(np.uint8)
    # observation descriptors: one pool row per observation
    M = len(window["obs_cam"])
    obs_pt = np.repeat(np.arange(P), np.diff(window["obs_ptr"]))
    desc_pool = flip_bits(rng, base[obs_pt])
    obs_desc = rng.permutation(M).astype(np.int32)        # rows are not in observation order
    pool = np.zeros_like(desc_pool)
    pool[obs_desc] = desc_pool
    kf_centers = window["cams_true"][:, 3:].astype(np.float32)
    eligible = (rng.random(P) < 0.9).astype(np.uint8)
    frame = dict(pose=pose.reshape(16), K=window["K"], width=W, height=H, keypoints=kp,
                 descriptors=desc, kp_matched=kp_matched)
    if kdtree_build is not None:
        node_kp, left, right, root = kdtree_build(kp)
        frame.update(kd_node_kp=node_kp, kd_left=left, kd_right=right, kd_root=root)
    mp = dict(positions=window["points"].astype(np.float32), eligible=eligible,
              obs_ptr=window["obs_ptr"], obs_kf=window["obs_cam"], obs_desc=obs_desc,
              kf_centers=kf_centers, desc_pool=pool)
    return frame, mp


def make_tracks(n_tracks=2000, n_frames=12, config_id=6, outlier_frac=0.05, noise_px=0.5, far_frac=0.3,
                max_sightings=10, image=(1920, 1080), K=(1000.0, 1000.0, 960.0, 540.0)):
    """Synthetic input of Mapper::triangulate_tracks (reference src/Mapper.cpp:222-305): a forward-moving,
    gently turning camera (`n_frames` trajectory poses, the last one is the key frame), `n_tracks` feature
    tracks, each sighted in a run of consecutive frames ending at the key frame.  `far_frac` of the landmarks
    are far away (low parallax: they exercise the requirement / quota top-up), `outlier_frac` of the tracks
    get one corrupted sighting (inconsistent)."""
    rng = np.random.default_rng(0x5EED0000 + config_id)
    fx, fy, cx, cy = K
    W, H = image
    poses = np.zeros((n_frames, 16), np.float32)
    for f in range(n_frames):
        yaw = np.deg2rad(0.4 * f)
        R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        c = np.array([0.03 * f, 0.0, 0.12 * f])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ c
        poses[f] = T.astype(np.float32).reshape(16)
    kf = n_frames - 1
    Tk = poses[kf].reshape(4, 4).astype(np.float64)
    # landmarks in the key frame's frustum
    depth = np.where(rng.random(n_tracks) < far_frac, rng.uniform(60, 400, n_tracks), rng.uniform(3, 25, n_tracks))
    u = rng.uniform(50, W - 50, n_tracks)
    v = rng.uniform(50, H - 50, n_tracks)
    Xc = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], 1)
    Xw = (Xc - Tk[:3, 3]) @ Tk[:3, :3]          # R^T (Xc - t)
    track_uv = np.zeros((n_tracks, 2), np.float32)
    sight_ptr = [0]
    sight_pose, sight_uv = [], []
    skip = (rng.random(n_tracks) < 0.03).astype(np.uint8)
    for t in range(n_tracks):
        ns = int(rng.integers(0, max_sightings + 1)) if rng.random() < 0.03 else int(rng.integers(2, max_sightings + 1))
        ns = min(ns, kf)
        frames = list(range(kf - ns, kf))           # sightings in the frames before the key frame
        bad = rng.random() < outlier_frac and ns >= 2
        bad_at = int(rng.integers(1, ns)) if bad else -1
        for j, f in enumerate(frames):
            T = poses[f].reshape(4, 4).astype(np.float64)
            pc = T[:3, :3] @ Xw[t] + T[:3, 3]
            px = np.array([fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy]) + rng.normal(0, noise_px, 2)
            if j == bad_at:
                px += rng.uniform(8, 30, 2) * rng.choice([-1, 1], 2)
            sight_pose.append(f)
            sight_uv.append(px)
        pk = Tk[:3, :3] @ Xw[t] + Tk[:3, 3]
        track_uv[t] = np.array([fx * pk[0] / pk[2] + cx, fy * pk[1] / pk[2] + cy]) + rng.normal(0, noise_px, 2)
        sight_ptr.append(len(sight_pose))
    return dict(track_uv=track_uv, skip=skip, sight_ptr=np.array(sight_ptr, np.int32),
                sight_pose=np.array(sight_pose, np.int32).reshape(-1),
                sight_uv=np.array(sight_uv, np.float32).reshape(-1, 2), poses=poses, kf_pose=kf,
                K=np.array(K, np.float32))


def make_pose_graph(n_kf=60, n_loops=3, laps=1.25, drift_rot=2e-3, drift_trans=2e-2, loop_noise=(2e-3, 1e-2),
                    outlier_loops=0, seed=0):
    """Key frames of a camera driving `laps` laps of an oval (y up, gravity = -y), odometry with accumulated drift, and
    loop constraints between key frames that see the same place one lap apart (reference LoopDetector -> pose_graph,
    src/Slam.cpp:258-268).  Returns dict(poses [n,4,4] f32 drifted world->camera, poses_true, loops [(from, to,
    relative 4x4 f64)], gravity): relative = T_from T_to^-1 measured on the TRUE trajectory plus noise, `from` the newer
    key frame as the detector emits them.  `outlier_loops` of them get a gross error (exercises the Huber loss)."""
    rng = np.random.default_rng(0x50600000 + seed)
    per_lap = int(round(n_kf / laps))
    T_true = []
    for i in range(n_kf):
        a = 2 * np.pi * i / per_lap
        centre = np.array([40.0 * np.cos(a), 0.3 * np.sin(3 * a), 25.0 * np.sin(a)])
        R_wc = yaw_matrix(-np.rad2deg(a)) @ rodrigues(np.array([0.02 * np.sin(2 * a), 0.0, 0.03 * np.cos(a)]))
        T_true.append(make_pose(R_wc, centre).astype(np.float64))
    T_true = np.stack(T_true)
    # odometry: true relative motion + noise, chained from the first true pose
    T = [T_true[0]]
    for i in range(1, n_kf):
        rel = T_true[i] @ np.linalg.inv(T_true[i - 1])               # T_i = rel T_{i-1}
        N = np.eye(4)
        N[:3, :3] = rodrigues(rng.normal(0, drift_rot, 3))
        N[:3, 3] = rng.normal(0, drift_trans, 3)
        T.append(N @ rel @ T[-1])
    poses = np.stack(T).astype(np.float32)
    loops = []
    cand = [i for i in range(per_lap, n_kf)]
    pick = rng.choice(cand, size=min(n_loops, len(cand)), replace=False) if cand else []
    for k, i in enumerate(sorted(int(v) for v in pick)):
        j = i - per_lap
        rel = T_true[i] @ np.linalg.inv(T_true[j])
        N = np.eye(4)
        N[:3, :3] = rodrigues(rng.normal(0, loop_noise[0], 3))
        N[:3, 3] = rng.normal(0, loop_noise[1], 3)
        if k < outlier_loops:
            N[:3, :3] = rodrigues(np.array([0.0, 0.4, 0.0]))
            N[:3, 3] = np.array([3.0, 0.0, -2.0])
        loops.append((i, j, N @ rel))
    return dict(poses=poses, poses_true=T_true, loops=loops, gravity=np.array([0.0, -9.80665, 0.0]))
