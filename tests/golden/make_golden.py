#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root).

The reference (GregVS/Racing-SLAM) ships no golden vectors for this path and
cannot be built here (OpenCV / Eigen / Ceres absent) — PARITY UNPINNED.  These
fixtures therefore pin the CPU oracle (oracle/*.c, validated against numpy /
scipy in tests/test_oracle_cpu.py): seeded synthetic inputs and the oracle's
outputs, small enough to commit.  tests/test_golden.py checks the oracle against
them on the CPU (regression pin) and the HIP library against them on the GPU.

    python tests/golden/make_golden.py            # everything
    python tests/golden/make_golden.py ba_trace   # only the sections whose name is given
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle as O  # noqa: E402

synth = importlib.import_module("racing-slam_amd.synth")
OUT = os.path.dirname(os.path.abspath(__file__))


TRACE_FIELDS = ("cost", "candidate_cost", "model_cost_change", "radius", "step_norm", "x_norm", "outcome")

# the LM-schedule fixtures: (name, make_ba_window arguments, max_num_iterations); chosen because the oracle
# REJECTS steps on them (outcomes A R R A ... / R A R A ...) and, with 60 iterations, terminates on the function tolerance
TRACE_CASES = (
    ("rej36", dict(n_kf=8, n_points=150, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0), 10),
    ("rej48", dict(n_kf=5, n_points=100, run_max=5, config_id=48, outlier_frac=0.15, rot_noise_deg=1.0), 10),
    ("conv36", dict(n_kf=8, n_points=150, run_max=5, config_id=36, outlier_frac=0.1, rot_noise_deg=2.0), 60),
    ("conv50", dict(n_kf=5, n_points=80, run_max=5, config_id=50, outlier_frac=0.0, pixel_noise=0.3), 60),
)


def trace_arrays(tr):
    return {k: np.array([t[k] for t in tr], np.int32 if k == "outcome" else np.float64) for k in TRACE_FIELDS}


def make_ba_trace():
    """Per-iteration records of the trust-region loop: the oracle's and, beside them, those of the independent
    dense numpy LM of tests/dense_lm.py (complex-step Jacobians, no Schur complement)."""
    import dense_lm as D
    blob = {}
    for name, kw, iters in TRACE_CASES:
        b = synth.make_ba_window(**kw)
        opt = O.default_options()
        opt.max_num_iterations = iters
        args = (b["cams"], b["cam_free"], b["points"], b["obs_ptr"], b["obs_cam"], b["obs_uv"], b["K"])
        cams, pts, s, tr = O.bundle_adjust_trace(*args, options=opt)
        x, s2, tr2 = D.solve(D.Problem(*args), max_iter=iters)
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in tr2], name
        for k in ("cams", "cam_free", "points", "obs_ptr", "obs_cam", "obs_uv", "K"):
            blob[f"{name}_{k}"] = b[k]
        blob[f"{name}_max_iter"] = np.int32(iters)
        blob[f"{name}_out_cams"], blob[f"{name}_out_points"] = cams, pts
        blob[f"{name}_summary"] = np.array([s["termination"], s["iterations"], s["successful_steps"], s["usable"]], np.int32)
        blob[f"{name}_costs"] = np.array([s["initial_cost"], s["final_cost"], s["final_radius"]])
        for k, v in trace_arrays(tr).items():
            blob[f"{name}_oracle_{k}"] = v
        for k, v in trace_arrays(tr2).items():
            blob[f"{name}_dense_{k}"] = v
        print(name, "".join({1: "A", 0: "R", -1: "I", 2: "T"}[t["outcome"]] for t in tr), "termination", s["termination"])
    np.savez_compressed(os.path.join(OUT, "bundle_adjust_trace.npz"), **blob)


IMU_KEYS = ("cam_i", "cam_j", "duration", "rotation", "velocity", "position", "covariance", "bias_gyro", "bias_accel",
            "bias_jacobian", "cam_velocity", "cam_bias", "gravity")


def imu_blob(prefix, imu):
    b = {f"{prefix}_{k}": np.asarray(imu[k]) for k in IMU_KEYS}
    b[f"{prefix}_sigmas"] = np.array([imu["gyro_bias_sigma"], imu["accel_bias_sigma"]])
    return b


def make_inertial():
    """§8(f) rank 2: the inertial residual blocks.  A 7-key-frame window with IMU factor pairs between consecutive free
    frames (one pair left out, like a gap in the stream), solved by the oracle (jets) and, beside it, by the dense numpy
    LM with complex-step Jacobians; refine_pose with a RotationPrior and with an InertialDelta."""
    import dense_lm as D
    blob = {}
    w = synth.make_ba_window(n_kf=7, n_points=150, run_max=5, config_id=62, outlier_frac=0.08, rot_noise_deg=1.5)
    imu = synth.make_imu(w, skip={(4, 5)})
    args = (w["cams"], w["cam_free"], w["points"], w["obs_ptr"], w["obs_cam"], w["obs_uv"], w["K"])
    c, p, v, b, s, tr = O.bundle_adjust_inertial(*args, imu, trace=True)
    prob = D.Problem(*args, imu=imu)
    x, s2, tr2 = D.solve(prob)
    assert [t["outcome"] for t in tr] == [t["outcome"] for t in tr2]
    for k in ("cams", "cam_free", "points", "obs_ptr", "obs_cam", "obs_uv", "K"):
        blob[f"ba_{k}"] = w[k]
    blob.update(imu_blob("ba_imu", imu))
    blob.update(ba_out_cams=c, ba_out_points=p, ba_out_velocity=v, ba_out_bias=b,
                ba_summary=np.array([s["termination"], s["iterations"], s["successful_steps"], s["usable"]], np.int32),
                ba_costs=np.array([s["initial_cost"], s["final_cost"], s["final_radius"]]))
    for k, a in trace_arrays(tr).items():
        blob[f"ba_oracle_{k}"] = a
    for k, a in trace_arrays(tr2).items():
        blob[f"ba_dense_{k}"] = a
    print("inertial BA", "".join({1: "A", 0: "R", -1: "I", 2: "T"}[t["outcome"]] for t in tr), s["final_cost"])
    # refine_pose forms on camera 3 of a 4-camera window
    w4 = synth.make_ba_window(n_kf=4, n_points=50, run_max=4, config_id=21, outlier_frac=0.05)
    sel = np.flatnonzero(w4["obs_cam"] == 3)
    obs_pt = np.repeat(np.arange(50), np.diff(w4["obs_ptr"]))[sel]
    pts, uv = w4["points_true"][obs_pt], w4["obs_uv"][sel]
    pred = synth.rodrigues(w4["cams_true"][3, :3]) @ synth.rodrigues(np.array([0.004, -0.003, 0.002]))
    cam1, _, s1 = O.refine_pose_inertial(w4["cams"][3], pts, uv, w4["K"], prior=(pred, 0.01))
    imu4 = synth.make_imu(w4)
    f = [i for i in range(len(imu4["cam_i"])) if imu4["cam_j"][i] == 3][0]
    one = {k: (np.asarray(imu4[k])[f:f + 1] if k in IMU_KEYS[:10] else imu4[k]) for k in imu4}
    d = dict(imu=one, prev_pose=w4["cams_true"][2], prev_velocity=imu4["cam_velocity_true"][2], prev_bias=imu4["cam_bias"][2],
             velocity=imu4["cam_velocity"][3])
    cam2, vel2, s2r = O.refine_pose_inertial(w4["cams"][3], pts, uv, w4["K"], delta=d)
    blob.update(rp_cam=w4["cams"][3], rp_points=pts, rp_uv=uv, rp_K=w4["K"], rp_predicted=pred, rp_sigma=np.float64(0.01),
                rp_prior_out_cam=cam1, rp_prior_summary=np.array([s1["termination"], s1["iterations"], s1["successful_steps"], s1["usable"]], np.int32),
                rp_prior_costs=np.array([s1["initial_cost"], s1["final_cost"]]),
                rp_prev_pose=d["prev_pose"], rp_prev_velocity=d["prev_velocity"], rp_prev_bias=d["prev_bias"], rp_velocity=d["velocity"],
                rp_delta_out_cam=cam2, rp_delta_out_velocity=vel2,
                rp_delta_summary=np.array([s2r["termination"], s2r["iterations"], s2r["successful_steps"], s2r["usable"]], np.int32),
                rp_delta_costs=np.array([s2r["initial_cost"], s2r["final_cost"]]))
    blob.update(imu_blob("rp_imu", one))
    np.savez_compressed(os.path.join(OUT, "inertial.npz"), **blob)


def make_pose_graph():
    """a14: 30 key frames / 4 loop constraints that contradict the odometry grossly (rejected first step, Huber branch),
    SE(3) and 4-DoF; outputs = the oracle's (pinned by tests/dense_lm.py in tests/test_pose_graph.py)."""
    g = synth.make_pose_graph(n_kf=30, n_loops=4, laps=1.4, seed=2, drift_rot=2e-2, drift_trans=0.3)
    loops = []
    for a, b, rel in g["loops"]:
        N = np.eye(4)
        N[:3, :3] = synth.rodrigues(np.array([0.3, 2.6, 0.2]))
        N[:3, 3] = [8.0, 1.0, -6.0]
        loops.append((a, b, N @ rel))
    blob = dict(poses=g["poses"].reshape(-1, 16), gravity=g["gravity"], loops_ft=np.array([(a, b) for a, b, _ in loops], np.int32),
                loops_rel=np.stack([r for _, _, r in loops]))
    for tag, fd in (("se3", False), ("dof4", True)):
        out, s, tr = O.pose_graph(g["poses"], loops, four_dof=fd, gravity=g["gravity"], trace=True)
        assert s["usable"] == 1 and 0 in [t["outcome"] for t in tr]
        blob[f"{tag}_poses"] = out.reshape(-1, 16)
        blob[f"{tag}_outcomes"] = np.array([t["outcome"] for t in tr], np.int32)
        blob[f"{tag}_costs"] = np.array([t["cost"] for t in tr])
        blob[f"{tag}_final_cost"] = np.float64(s["final_cost"])
    np.savez_compressed(os.path.join(OUT, "pose_graph.npz"), **blob)


def main():
    O.build()
    only = set(sys.argv[1:])
    if not only or "pose_graph" in only:
        make_pose_graph()
    if not only or "ba_trace" in only:
        make_ba_trace()
    if not only or "inertial" in only:
        make_inertial()
    if only and only != {"all"}:
        return
    # -- a4: descriptor matching, 96 x 80 rows incl. exact ties and threshold boundaries
    rng = np.random.default_rng(20261004)
    t = rng.integers(0, 256, (80, 32), dtype=np.uint8)
    q = np.zeros((96, 32), np.uint8)
    for i in range(96):
        bits = np.unpackbits(t[i % 80])
        flip = rng.choice(256, size=[0, 5, 20, 48, 64, 65, 90, 128][i % 8], replace=False)
        bits[flip] ^= 1
        q[i] = np.packbits(bits)
    t[40:44] = t[3]
    i0, d0, i1, d1 = O.hamming_knn2(q, t)
    mq, mt = O.match_descriptors(q, t)
    np.savez_compressed(os.path.join(OUT, "match_descriptors.npz"), query=q, train=t, idx0=i0, dist0=d0, idx1=i1,
                        dist1=d1, match_query=mq, match_train=mt)

    # -- a6: triangulation, 64 correspondences of the cfg-1 pair (both gate settings)
    pr = synth.make_pair(1)
    a1, a2, ns = pr["truth12"]
    uv1, uv2 = pr["kp1"][a1[:64]], pr["kp2"][a2[:64]]
    uv2[5] += 40.0          # a bad correspondence
    d = O.triangulate(uv1, uv2, pr["poses"], pr["K"])
    m = O.triangulate(uv1, uv2, pr["poses"], pr["K"], min_parallax_cosine=1.0, max_reproj=4.0)
    np.savez_compressed(os.path.join(OUT, "triangulate.npz"), uv1=uv1, uv2=uv2, poses=pr["poses"], K=pr["K"],
                        xyz=d["xyz"], keep_default=d["keep"], out_index_default=d["out_index"],
                        keep_mapper=m["keep"], out_index_mapper=m["out_index"])

    # -- §8(f)1: Mapper::triangulate_tracks body, 160 tracks over 10 frames; quota 24 so that the top-up runs
    tk = synth.make_tracks(n_tracks=160, n_frames=10, config_id=8, far_frac=0.85)
    tr = O.triangulate_tracks(tk["track_uv"], tk["sight_ptr"], tk["sight_pose"], tk["sight_uv"], tk["poses"],
                              tk["kf_pose"], tk["K"], skip=tk["skip"], min_new_points=24)
    np.savez_compressed(os.path.join(OUT, "triangulate_tracks.npz"), **{k: tk[k] for k in
                        ("track_uv", "skip", "sight_ptr", "sight_pose", "sight_uv", "poses", "K")},
                        kf_pose=np.int32(tk["kf_pose"]), min_new_points=np.int32(24), status=tr["status"], xyz=tr["xyz"],
                        parallax_cos=tr["parallax_cos"], required_cos=tr["required_cos"], accepted=tr["accepted"],
                        n_topped_up=np.int32(tr["n_topped_up"]), inconsistent=tr["inconsistent"])

    # -- a2: reprojection-gated matching, 5 keyframes / 150 landmarks / 300 keypoints
    w = synth.make_ba_window(n_kf=5, n_points=150, run_max=4)
    frame, mp = synth.make_match_scene(w, n_keypoints=300, kdtree_build=O.kdtree_build)
    r0 = O.reproj_match(frame, mp, replace=0)
    r1 = O.reproj_match(frame, mp, replace=1)
    blob = {f"frame_{k}": np.asarray(v) for k, v in frame.items()}
    blob.update({f"map_{k}": np.asarray(v) for k, v in mp.items()})
    for tag, r in (("r0", r0), ("r1", r1)):
        blob.update({f"{tag}_{k}": v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "reproj_match.npz"), **blob)

    # -- a12: 4-camera / 50-landmark bundle adjustment, 10 LM iterations
    b = synth.make_ba_window(n_kf=4, n_points=50, run_max=4, config_id=21, outlier_frac=0.05)
    cams, pts, s = O.bundle_adjust(b["cams"], b["cam_free"], b["points"], b["obs_ptr"], b["obs_cam"], b["obs_uv"], b["K"])
    np.savez_compressed(os.path.join(OUT, "bundle_adjust.npz"), cams=b["cams"], cam_free=b["cam_free"], points=b["points"],
                        obs_ptr=b["obs_ptr"], obs_cam=b["obs_cam"], obs_uv=b["obs_uv"], K=b["K"], out_cams=cams,
                        out_points=pts, summary=np.array([s["termination"], s["iterations"], s["successful_steps"],
                                                          s["usable"]], np.int32),
                        costs=np.array([s["initial_cost"], s["final_cost"], s["final_radius"]]))

    # -- a13: pose-only refinement on 120 observations
    sel = np.flatnonzero(b["obs_cam"] == 3)
    obs_pt = np.repeat(np.arange(50), np.diff(b["obs_ptr"]))[sel]
    rp_pts = b["points_true"][obs_pt]
    cam, rs_ = O.refine_pose(b["cams"][3], rp_pts, b["obs_uv"][sel], b["K"])
    np.savez_compressed(os.path.join(OUT, "refine_pose.npz"), cam=b["cams"][3], points=rp_pts, uv=b["obs_uv"][sel], K=b["K"],
                        out_cam=cam, summary=np.array([rs_["termination"], rs_["iterations"], rs_["successful_steps"],
                                                       rs_["usable"]], np.int32),
                        costs=np.array([rs_["initial_cost"], rs_["final_cost"]]))

    # -- a10: pose packing
    poses = np.stack([synth.make_pose(synth.rodrigues(rng.normal(size=3) * 0.7).T, rng.normal(size=3) * 3) for _ in range(16)])
    packed = np.stack([O.pack_pose(p) for p in poses])
    unpacked = np.stack([O.unpack_pose(c) for c in packed])
    np.savez_compressed(os.path.join(OUT, "pack_pose.npz"), poses=poses, packed=packed, unpacked=unpacked)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
