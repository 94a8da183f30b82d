"""§8 a14: optimization::pose_graph (reference src/Optimization.cpp:376-639).

PARITY UNPINNED by the reference (no fixtures upstream).  Three layers, none sharing code:
  oracle/pose_graph.c   jets + DENSE normal equations, natural ordering               (the checker)
  tests/dense_lm.py     complex-step Jacobians + numpy dense solve                    (pins the checker)
  csrc/pose_graph.cpp   C++ dual numbers + block-envelope Cholesky after an RCM ordering  (the product, host function)
plus the device side of a loop closure: K14 rs_transform_points and rs_map_pose_graph (gpu-marked).
"""
import numpy as np
import pytest

import dense_lm as D

CODE = {1: "A", 0: "R", -1: "I", 2: "T"}


def gross_loops(synth, g):
    """every loop constraint grossly inconsistent with the odometry: forces rejected steps and the Huber branch"""
    out = []
    for a, b, rel in g["loops"]:
        N = np.eye(4)
        N[:3, :3] = synth.rodrigues(np.array([0.3, 2.6, 0.2]))
        N[:3, 3] = [8.0, 1.0, -6.0]
        out.append((a, b, N @ rel))
    return out


def cases(synth):
    g0 = synth.make_pose_graph(n_kf=24, n_loops=3, laps=1.3, seed=0)
    g1 = synth.make_pose_graph(n_kf=26, n_loops=4, laps=1.4, outlier_loops=2, seed=1)
    g2 = synth.make_pose_graph(n_kf=30, n_loops=4, laps=1.4, seed=2, drift_rot=2e-2, drift_trans=0.3)
    g2 = dict(g2, loops=gross_loops(synth, g2))
    return [("clean", g0), ("outliers", g1), ("gross", g2)]


def up_of(g):
    return -g["gravity"] / np.linalg.norm(g["gravity"])


# ------------------------------------------------------------------------------------------- the oracle, pinned
@pytest.mark.parametrize("four_dof", [False, True])
def test_oracle_edge_jacobian_vs_central_differences(oracle, synth, four_dof):
    rng = np.random.default_rng(5)
    g = synth.make_pose_graph(n_kf=12, n_loops=1, laps=1.0, seed=4)
    P = g["poses"]
    for trial in range(6):
        a, b = rng.choice(len(P), 2, replace=False)
        xa, xb = oracle.pack_pose(P[a]), oracle.pack_pose(P[b])
        if four_dof:
            xa, xb = np.concatenate([[0.03 * trial], xa[3:]]), np.concatenate([[-0.02 * trial], xb[3:]])
        rel = oracle.pose_relative(P[a], P[b]) @ np.block([[synth.rodrigues(rng.normal(0, 0.05, 3)), rng.normal(0, 0.2, (3, 1))], [np.zeros((1, 3)), np.ones((1, 1))]])
        R0a, R0b = P[a][:3, :3].astype(np.float64), P[b][:3, :3].astype(np.float64)
        r, J = oracle.pose_graph_edge(four_dof, xa, xb, R0a, R0b, up_of(g), rel, trial % 2)
        bs = 4 if four_dof else 6
        num = np.zeros((6, 2 * bs))
        for k in range(2 * bs):
            h = 1e-6
            xp, xm = np.concatenate([xa, xb]), np.concatenate([xa, xb])
            xp[k] += h
            xm[k] -= h
            rp, _ = oracle.pose_graph_edge(four_dof, xp[:bs], xp[bs:], R0a, R0b, up_of(g), rel, trial % 2)
            rm, _ = oracle.pose_graph_edge(four_dof, xm[:bs], xm[bs:], R0a, R0b, up_of(g), rel, trial % 2)
            num[:, k] = (rp - rm) / (2 * h)
        assert np.allclose(J[:, :2 * bs], num, rtol=1e-6, atol=1e-6)
        assert np.all(J[:, 2 * bs:] == 0)


def test_oracle_pose_relative_vs_numpy(oracle, synth):
    g = synth.make_pose_graph(n_kf=20, n_loops=1, seed=9)
    P = g["poses"]
    for i in range(len(P) - 1):
        ref = P[i].astype(np.float64) @ np.linalg.inv(P[i + 1].astype(np.float64))
        assert np.allclose(oracle.pose_relative(P[i], P[i + 1]), ref, atol=3e-5)      # an f32 inverse of poses 40 m out


@pytest.mark.parametrize("four_dof", [False, True])
def test_oracle_trajectory_vs_independent_dense_lm(oracle, synth, four_dof):
    """whole LM trajectory of the oracle == dense numpy LM with complex-step Jacobians (1e-8 on every record)"""
    seen = ""
    for name, g in cases(synth):
        P = g["poses"]
        out, s, tr = oracle.pose_graph(P, g["loops"], four_dof=four_dof, gravity=g["gravity"], trace=True)
        x0 = np.stack([oracle.pack_pose(p) for p in P])
        if four_dof:
            x0 = np.concatenate([np.zeros((len(P), 1)), x0[:, 3:]], axis=1)
        seq = [oracle.pose_relative(P[i], P[i + 1]) for i in range(len(P) - 1)]
        prob = D.PoseGraphProblem(P, x0, g["loops"], four_dof, up_of(g), seq_relative=seq)
        xb, s2, tr2 = D.solve(prob, max_iter=20)
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in tr2], name
        assert (s["termination"], s["iterations"], s["successful_steps"], s["usable"]) == \
               (s2["termination"], s2["iterations"], s2["successful_steps"], s2["usable"])
        for a, b in zip(tr, tr2):
            for k in ("cost", "candidate_cost", "model_cost_change", "radius", "step_norm", "x_norm"):
                assert a[k] == pytest.approx(b[k], rel=1e-7, abs=1e-12), (name, k)
        assert s["final_cost"] == pytest.approx(s2["final_cost"], rel=1e-9)
        assert s["final_cost"] < s["initial_cost"]
        seen += "".join(CODE[t["outcome"]] for t in tr)
        # the optimum is written back through f32: compare the centres / rotations of the dense solution
        X = np.concatenate([x0[:1], xb.reshape(len(P) - 1, -1)])
        for i in range(len(P)):
            if four_dof:
                R = P[i][:3, :3].astype(np.float64) @ D.aa_to_matrix(-up_of(g) * X[i, 0])
                c = X[i, 1:]
            else:
                R, c = D.aa_to_matrix(X[i, :3]), X[i, 3:]
            assert np.allclose(out[i][:3, :3], R, atol=2e-6)
            assert np.allclose(out[i][:3, 3], -R @ c, atol=2e-4)
    assert "R" in seen and "A" in seen and "T" in seen      # the cases cover accepted, rejected and terminating steps


def test_oracle_transform_points_vs_numpy(oracle):
    rng = np.random.default_rng(2)
    n_kf, P = 9, 200
    deg = rng.integers(0, 5, P)
    obs_ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    obs_kf = np.concatenate([rng.choice(n_kf, d, replace=False) for d in deg] + [np.zeros(0, int)]).astype(np.int32)
    before = np.stack([rand_pose(rng) for _ in range(n_kf)])
    after = np.stack([rand_pose(rng) for _ in range(n_kf)])
    pos = rng.normal(0, 10, (P, 3)).astype(np.float32)
    got = oracle.transform_points(obs_ptr, obs_kf, before.reshape(-1, 16), after.reshape(-1, 16), pos)
    for p in range(P):
        if deg[p] == 0:
            assert np.array_equal(got[p], pos[p])
            continue
        o = obs_kf[obs_ptr[p]:obs_ptr[p + 1]].min()
        B, A = before[o].astype(np.float64), after[o].astype(np.float64)
        ref = A[:3, :3].T @ (B[:3, :3] @ pos[p] + B[:3, 3] - A[:3, 3])
        assert np.allclose(got[p], ref, atol=2e-5)


def rand_pose(rng):
    from scipy.spatial.transform import Rotation
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = Rotation.from_rotvec(rng.normal(0, 0.6, 3)).as_matrix().astype(np.float32)
    T[:3, 3] = rng.normal(0, 5, 3).astype(np.float32)
    return T


# ------------------------------------------------------------------------------------------- the product (host)
@pytest.mark.parametrize("four_dof", [False, True])
def test_product_matches_oracle(oracle, rs, synth, four_dof):
    """librsgpu's rs_pose_graph (sparse envelope solve, RCM ordering, C++ duals) against the dense oracle: identical
    schedule, records to 1e-7, corrected poses BIT-exact (they are rounded to f32 on write-back)."""
    big = synth.make_pose_graph(n_kf=150, n_loops=50, laps=2.3, outlier_loops=2, seed=3)
    for name, g in cases(synth) + [("laps", big)]:
        out, s, tr = oracle.pose_graph(g["poses"], g["loops"], four_dof=four_dof, gravity=g["gravity"], trace=True)
        got, rot, s2, tr2 = rs.pose_graph(g["poses"], g["loops"], four_dof=four_dof, gravity=g["gravity"])
        assert [t["outcome"] for t in tr] == [t["outcome"] for t in tr2], name
        for k in ("termination", "iterations", "successful_steps", "usable"):
            assert s[k] == s2[k], (name, k)
        for a, b in zip(tr, tr2):
            for k in ("cost", "candidate_cost", "model_cost_change", "radius", "step_norm", "x_norm"):
                assert a[k] == pytest.approx(b[k], rel=1e-7, abs=1e-12), (name, k)
        assert s2["final_cost"] == pytest.approx(s["final_cost"], rel=1e-10)
        assert s2["usable"] == 1
        assert np.array_equal(got, out), name
        # R_delta of apply_corrected_pose (:505): new^T old, f32
        ref = np.einsum("nki,nkj->nij", got[:, :3, :3].astype(np.float64), g["poses"][:, :3, :3].astype(np.float64))
        assert np.allclose(rot, ref, atol=5e-7)
        assert np.array_equal(got[0], g["poses"][0]) or np.allclose(got[0], g["poses"][0], atol=1e-6)   # first key frame constant


def test_product_pose_relative_is_the_oracles(oracle, rs, synth):
    g = synth.make_pose_graph(n_kf=40, n_loops=1, seed=6)
    P = g["poses"]
    for i in range(len(P) - 1):
        assert np.array_equal(rs.pose_relative(P[i], P[i + 1]), oracle.pose_relative(P[i], P[i + 1]))


def test_product_reference_early_outs(rs, oracle, synth):
    g = synth.make_pose_graph(n_kf=24, n_loops=3, laps=1.3, seed=0)
    P = g["poses"]
    # fewer than three key frames / no loops: "return false" (:546-548) — nothing changes, usable == 0
    for poses, loops in [(P[:2], g["loops"][:1]), (P, [])]:
        got, rot, s, tr = rs.pose_graph(poses, loops)
        assert s["usable"] == 0 and s["iterations"] == 0 and tr == []
        assert np.array_equal(got, poses)
        assert np.array_equal(rot, np.broadcast_to(np.eye(3, dtype=np.float32), rot.shape))
    # constraints out of range or from == to are skipped (:590-592); only skipped ones == the chain alone: the optimum is the input
    bad = [(5, 5, np.eye(4)), (99, 2, np.eye(4)), (-1, 3, np.eye(4))]
    got, rot, s, tr = rs.pose_graph(P, bad)
    o_out, o_s = oracle.pose_graph(P, bad)
    assert (s["termination"], s["iterations"], s["usable"]) == (o_s["termination"], o_s["iterations"], o_s["usable"])
    assert np.array_equal(got, o_out)
    assert np.abs(got - P).max() < 1e-5
    # four_dof without gravity falls back to SE(3) (:552-553)
    a = rs.pose_graph(P, g["loops"], four_dof=True, gravity=(0.0, 0.0, 1e-4))
    b = rs.pose_graph(P, g["loops"], four_dof=False)
    assert np.array_equal(a[0], b[0]) and a[2] == b[2]
    # the 4-DoF solve leaves roll / pitch alone: R_new = R0 * rot(up, yaw)  =>  R0^T R_new fixes `up`
    got4, _, s4, _ = rs.pose_graph(P, g["loops"], four_dof=True, gravity=g["gravity"])
    up = up_of(g)
    for i in range(len(P)):
        Rd = P[i][:3, :3].astype(np.float64).T @ got4[i][:3, :3].astype(np.float64)
        assert np.allclose(Rd @ up, up, atol=1e-5)
    assert s4["usable"] == 1


def test_product_improves_the_trajectory(rs, synth):
    g = synth.make_pose_graph(n_kf=300, n_loops=120, laps=2.4, seed=8)
    got, rot, s, tr = rs.pose_graph(g["poses"], g["loops"])

    def centres(T):
        T = np.asarray(T, np.float64)
        return -np.einsum("nji,nj->ni", T[:, :3, :3], T[:, :3, 3])
    e0 = np.linalg.norm(centres(g["poses"]) - centres(g["poses_true"]), axis=1)
    e1 = np.linalg.norm(centres(got) - centres(g["poses_true"]), axis=1)
    assert s["usable"] == 1 and s["final_cost"] < 0.2 * s["initial_cost"]
    assert e1[150:].mean() < 0.6 * e0[150:].mean()


def test_golden_pose_graph(rs, oracle):
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "pose_graph.npz")
    z = np.load(path)
    for tag in ("se3", "dof4"):
        loops = [(int(a), int(b), rel) for (a, b), rel in zip(z["loops_ft"], z["loops_rel"])]
        for impl in ("oracle", "product"):
            if impl == "oracle":
                out, s, tr = oracle.pose_graph(z["poses"], loops, four_dof=(tag == "dof4"), gravity=z["gravity"], trace=True)
            else:
                out, _, s, tr = rs.pose_graph(z["poses"], loops, four_dof=(tag == "dof4"), gravity=z["gravity"])
            assert np.array_equal(out.reshape(-1, 16), z[f"{tag}_poses"]), (tag, impl)
            assert [t["outcome"] for t in tr] == list(z[f"{tag}_outcomes"])
            assert np.allclose([t["cost"] for t in tr], z[f"{tag}_costs"], rtol=1e-9)
            assert s["final_cost"] == pytest.approx(float(z[f"{tag}_final_cost"]), rel=1e-9)


# ------------------------------------------------------------------------------------------- the device side
def flat_map(rng, n_kf, P):
    deg = rng.integers(0, 6, P)
    obs_ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    obs_kf = np.concatenate([rng.choice(n_kf, d, replace=False) for d in deg] + [np.zeros(0, int)]).astype(np.int32)
    return obs_ptr, obs_kf


@pytest.mark.gpu
def test_transform_points_bit_exact(ctx, rs, oracle, synth):
    import torch
    rng = np.random.default_rng(12)
    g = synth.make_pose_graph(n_kf=80, n_loops=20, laps=1.6, seed=5)
    after, _, s, _ = rs.pose_graph(g["poses"], g["loops"])
    assert s["usable"] == 1
    P = 50_000
    obs_ptr, obs_kf = flat_map(rng, 80, P)
    obs_kf[rng.integers(0, len(obs_kf), 50)] = -1                 # owners outside the list: the point stays (:524-527)
    pos = rng.normal(0, 30, (P, 3)).astype(np.float32)
    want = oracle.transform_points(obs_ptr, obs_kf, g["poses"].reshape(-1, 16), after.reshape(-1, 16), pos)
    dev = torch.device("cuda:0")
    d_pos = torch.from_numpy(pos.copy()).to(dev)
    ctx.transform_points(torch.from_numpy(obs_ptr).to(dev), torch.from_numpy(obs_kf).to(dev),
                         torch.from_numpy(g["poses"].reshape(-1, 16).copy()).to(dev), torch.from_numpy(after.reshape(-1, 16).copy()).to(dev), d_pos)
    ctx.synchronize()
    got = d_pos.cpu().numpy()
    assert np.array_equal(got, want)
    moved = np.any(got != pos, axis=1)
    assert moved.sum() > 0.5 * P and (~moved).sum() > 0          # points without observations did not move


@pytest.mark.gpu
def test_resident_map_pose_graph(ctx, rs, oracle, synth):
    """rs_map_pose_graph == rs_pose_graph on the mirror's poses + transform_points on the mirror's topology; the map
    keeps matching / optimising from the corrected state (positions and centres refreshed)."""
    rng = np.random.default_rng(3)
    n_kf, P = 40, 3000
    g = synth.make_pose_graph(n_kf=n_kf, n_loops=10, laps=1.5, seed=7)
    m = rs.ResidentMap(ctx)
    frames = []
    for k in range(n_kf):
        kp = rng.uniform(0, 600, (8, 2)).astype(np.float32)
        desc = rng.integers(0, 256, (8, 32), dtype=np.uint8)
        f = rs.ResidentFrame(ctx, kp, desc)
        frames.append(f)
        assert m.add_keyframe(f, g["poses"][k]) == k
    obs_ptr, obs_kf = flat_map(rng, n_kf, P)
    pos = rng.normal(0, 30, (P, 3)).astype(np.float32)
    used = {}
    keep_ptr, keep_kf = [0], []
    for p in range(P):
        assert m.add_point(pos[p]) == p
        for kf in obs_kf[obs_ptr[p]:obs_ptr[p + 1]]:
            slot = used.get(int(kf), 0)
            if slot < 8:                                  # 8 keypoints per synthetic key frame
                m.add_observation(p, int(kf), slot)
                used[int(kf)] = slot + 1
                keep_kf.append(int(kf))
        keep_ptr.append(len(keep_kf))
    want_poses, want_rot, s0, _ = rs.pose_graph(g["poses"], g["loops"], four_dof=True, gravity=g["gravity"])
    want_pos = oracle.transform_points(np.array(keep_ptr, np.int32), np.array(keep_kf, np.int32), g["poses"].reshape(-1, 16),
                                       want_poses.reshape(-1, 16), pos)
    s, poses, rot = m.pose_graph(g["loops"], four_dof=True, gravity=g["gravity"])
    assert s == s0 and s["usable"] == 1
    assert np.array_equal(poses, want_poses) and np.array_equal(rot, want_rot)
    got_pos = m.positions()
    assert np.array_equal(got_pos, want_pos)
    assert np.any(got_pos != pos)
    # rejected solve: nothing changes
    s2, poses2, _ = m.pose_graph([], four_dof=False)
    assert s2["usable"] == 0 and np.array_equal(poses2, want_poses) and np.array_equal(m.positions(), want_pos)
    m.close()
    for f in frames:
        f.close()
