"""Independent dense restatement of the trust-region loop, used to pin the LM SCHEDULE of the oracle.

Nothing here shares code or formulation with oracle/ba.c or the HIP kernels:
  * Jacobians by COMPLEX-STEP differentiation of the residual (machine precision, no jets,
    no analytic SO(3) Jacobian);
  * NO Schur complement: the full damped normal equations over all camera and point
    parameters are solved densely with numpy;
  * the rules of Ceres' TrustRegionMinimizer / LevenbergMarquardtStrategy as documented for
    Ceres 2.x defaults (the reference only sets the iteration cap and the
    linear solver, src/Optimization.cpp:127-134): Jacobi scaling 1/(1+|col|) from the first
    Jacobian, D^2 = clamp(|col|^2, 1e-6, 1e32)/radius on the scaled Jacobian, model cost change
    -(J s).(r + J s/2), parameter- and function-tolerance tests before acceptance,
    rho > 1e-3 accepts and radius /= max(1/3, 1-(2 rho-1)^3), rejection divides the radius by a
    factor that doubles.
The residual is the reference functor (src/Optimization.cpp:35-52) with ceres::HuberLoss
(a = sqrt(5.991), :311) applied through Ceres' corrector (rho'' <= 0: scale r and J by sqrt(rho')).
Small problems only (dense (6C+3P)^2 system).  The inertial residual blocks of the reference (IMU preintegration +
bias random walk in bundle_adjust, RotationPrior / InertialDelta in refine_pose) are covered the same way: numpy
restatements differentiated by complex step.
"""
import numpy as np

EPS = 2.220446049250313e-16


def rotate(aa, v):
    """ceres::AngleAxisRotatePoint for arrays [..., 3] (complex-capable)."""
    th2 = np.sum(aa * aa, axis=-1, keepdims=True)
    big = np.real(th2) > EPS
    th = np.sqrt(np.where(big, th2, 1.0))
    w = aa / th
    c, s = np.cos(th), np.sin(th)
    wxv = np.cross(w, v)
    dot = np.sum(w * v, axis=-1, keepdims=True)
    full = v * c + wxv * s + w * dot * (1.0 - c)
    small = v + np.cross(aa, v)
    return np.where(big, full, small)


def residuals(cams, pts, obs_cam, obs_pt, obs_uv, K):
    """[M, 2] reprojection residuals (src/Optimization.cpp:40-49)."""
    cam = cams[obs_cam]
    p = rotate(cam[:, :3], pts[obs_pt] - cam[:, 3:])
    fx, fy, cx, cy = [float(np.float32(k)) for k in K]
    uv = obs_uv.astype(np.float64)
    return np.stack([fx * p[:, 0] / p[:, 2] + cx - uv[:, 0], fy * p[:, 1] / p[:, 2] + cy - uv[:, 1]], axis=1)


def jacobian_blocks(cams, pts, obs_cam, obs_pt, obs_uv, K, h=1e-30):
    """Complex-step derivative of every residual w.r.t. its own camera (6) and point (3)."""
    M = len(obs_cam)
    jc = np.zeros((M, 2, 6))
    jp = np.zeros((M, 2, 3))
    for k in range(6):
        cz = cams.astype(np.complex128)
        cz[:, k] += 1j * h
        jc[:, :, k] = np.imag(residuals(cz, pts.astype(np.complex128), obs_cam, obs_pt, obs_uv, K)) / h
    for k in range(3):
        pz = pts.astype(np.complex128)
        pz[:, k] += 1j * h
        jp[:, :, k] = np.imag(residuals(cams.astype(np.complex128), pz, obs_cam, obs_pt, obs_uv, K)) / h
    return jc, jp


def aa_to_matrix(aa):
    """ceres::AngleAxisToRotationMatrix for ONE angle-axis vector (complex-capable), as a 3x3 array."""
    th2 = aa @ aa
    if np.real(th2) > EPS:
        th = np.sqrt(th2)
        w = aa / th
        c, s = np.cos(th), np.sin(th)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=aa.dtype)
        return c * np.eye(3) + s * K + (1 - c) * np.outer(w, w)
    return np.array([[1, -aa[2], aa[1]], [aa[2], 1, -aa[0]], [-aa[1], aa[0], 1]], dtype=aa.dtype)


def matrix_to_aa(R):
    """ceres::RotationMatrixToAngleAxis (through the quaternion; branches decided on real parts)."""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4, dtype=R.dtype)
    if np.real(tr) >= 0:
        t = np.sqrt(tr + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1], q[2], q[3] = (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if np.real(R[1, 1]) > np.real(R[0, 0]):
            i = 1
        if np.real(R[2, 2]) > np.real(R[i, i]):
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i + 1] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[j + 1] = (R[j, i] + R[i, j]) * t
        q[k + 1] = (R[k, i] + R[i, k]) * t
    s2 = q[1:] @ q[1:]
    if np.real(s2) > 0:
        s = np.sqrt(s2)
        # complex-capable atan2(s, c) for |angle| < pi: 2 atan(s / (sqrt(s^2 + c^2) + c)); the reference's own
        # cos < 0 branch (atan2(-s, -c)) is the same angle minus pi, handled by the sign flip below
        c = q[0]
        if np.real(c) < 0:
            s_, c_ = -s, -c
        else:
            s_, c_ = s, c
        two_theta = 2.0 * 2.0 * np.arctan(s_ / (np.sqrt(s_ * s_ + c_ * c_) + c_))
        return q[1:] * (two_theta / s)
    return q[1:] * 2.0


def imu_preintegration(f, gravity, pose_i, vel_i, bias_i, pose_j, vel_j):
    """PreintegrationError (reference src/ImuFactor.cpp:27-81) for one factor dict (row-major fields), whitened;
    complex-capable in every argument."""
    dt = np.result_type(pose_i, vel_i, bias_i, pose_j, vel_j)
    Ri, Rj = aa_to_matrix(pose_i[:3].astype(dt)), aa_to_matrix(pose_j[:3].astype(dt))
    db = np.concatenate([bias_i[:3] - f["bias_gyro"], bias_i[3:] - f["bias_accel"]])
    corr = f["bias_jacobian"].reshape(9, 6) @ db
    Rm = f["rotation"].reshape(3, 3) @ aa_to_matrix(corr[:3].astype(dt))
    mvel, mpos = f["velocity"] + corr[3:6], f["position"] + corr[6:9]
    T = f["duration"]
    Rs = Ri @ Rj.T
    sv = Ri @ (vel_j - vel_i - gravity * T)
    sp = Ri @ (pose_j[3:] - pose_i[3:] - vel_i * T - 0.5 * gravity * T * T)
    res = np.concatenate([matrix_to_aa(Rm.T @ Rs), sv - mvel, sp - mpos])
    L = np.linalg.cholesky(f["covariance"].reshape(9, 9))
    return np.linalg.solve(L, res)


def huber(s, a):
    b = a * a
    out = s > b
    r = np.sqrt(np.where(out, s, 1.0))
    rho = np.where(out, 2.0 * a * r - b, s)
    rho1 = np.where(out, a / r, 1.0)
    return rho, rho1


def reduced_system_condition(w, cams, pts, radius):
    """Condition of the Jacobi-scaled reduced camera system (point blocks eliminated) of a window at the state (cams, pts):
    {"undamped": (cond, smallest, largest eigenvalue), "final radius": ...} — what bounds how far two correct f64 solves of
    the window may differ (tools/tol_check.py, tests/test_gpu_round3.py::test_bundle_adjust_banded_reduced_solve)."""
    obs_pt = np.repeat(np.arange(len(pts)), np.diff(w["obs_ptr"]))
    obs_cam = np.asarray(w["obs_cam"], np.int64)
    r = residuals(cams, pts, obs_cam, obs_pt, np.asarray(w["obs_uv"], np.float32), w["K"])
    jc, jp = jacobian_blocks(cams, pts, obs_cam, obs_pt, np.asarray(w["obs_uv"], np.float32), w["K"])
    _, rho1 = huber(np.sum(r * r, axis=1), np.sqrt(5.991))
    free = np.asarray(w["cam_free"], bool)
    slot = np.where(free, np.cumsum(free) - 1, -1)
    n = 6 * int(free.sum())
    U = np.zeros((n, n))
    P = len(pts)
    V = np.zeros((P, 3, 3))
    Wl = [[] for _ in range(P)]
    for o in range(len(obs_cam)):
        wgt = rho1[o]
        p = obs_pt[o]
        V[p] += wgt * jp[o].T @ jp[o]
        s = slot[obs_cam[o]]
        if s >= 0:
            U[6 * s:6 * s + 6, 6 * s:6 * s + 6] += wgt * jc[o].T @ jc[o]
            Wl[p].append((s, wgt * jc[o].T @ jp[o]))
    out = {}
    for name, rad in (("undamped", None), ("final radius", radius)):
        # Jacobi scaling 1 / (1 + sqrt(diag)) and Ceres' damping diag / radius
        sc = 1.0 / (1.0 + np.sqrt(np.diag(U)))
        S = U.copy()
        if rad is not None:
            S += np.diag(np.clip(np.diag(U) * sc * sc, 1e-6, 1e32) / (rad * sc * sc))
        for p in range(P):
            Vp = V[p].copy()
            if rad is not None:
                sp = 1.0 / (1.0 + np.sqrt(np.diag(V[p])))
                Vp += np.diag(np.clip(np.diag(V[p]) * sp * sp, 1e-6, 1e32) / (rad * sp * sp))
            Vi = np.linalg.inv(Vp)
            for (s1, W1) in Wl[p]:
                for (s2, W2) in Wl[p]:
                    S[6 * s1:6 * s1 + 6, 6 * s2:6 * s2 + 6] -= W1 @ Vi @ W2.T
        Ss = S * sc[:, None] * sc[None, :]
        ev = np.linalg.eigvalsh(0.5 * (Ss + Ss.T))
        out[name] = (ev[-1] / max(ev[0], 1e-300), ev[0], ev[-1])
    return out



class Problem:
    def __init__(self, cams, cam_free, pts, obs_ptr, obs_cam, obs_uv, K, huber_a=np.sqrt(5.991), imu=None,
                 points_constant=False, prior=None, delta=None):
        """imu: synth.make_imu dict (bundle_adjust with IMU factor pairs).  refine_pose forms: points_constant=True
        with prior=(predicted 3x3, sigma) (PredictedRotationError, src/Optimization.cpp:75-94) or
        delta=dict(imu=one-factor dict, prev_pose, prev_velocity, prev_bias, velocity) (src/Optimization.cpp:237-251)."""
        self.points_constant, self.prior, self.delta = points_constant, prior, delta
        self.cams0 = np.array(cams, np.float64)
        self.pts0 = np.array(pts, np.float64)
        self.obs_cam = np.asarray(obs_cam, np.int64)
        self.obs_pt = np.repeat(np.arange(len(pts)), np.diff(obs_ptr)).astype(np.int64)
        self.obs_uv = np.asarray(obs_uv, np.float32)
        self.K = K
        self.a = float(huber_a)
        # parameter blocks of the reduced program: free cameras that carry a residual, then all points
        seen = np.zeros(len(cams), bool)
        seen[self.obs_cam] = True
        # inertial factor pairs (src/Optimization.cpp:317-346): velocity (3) + bias (6) blocks of the frames they touch
        self.imu = imu
        self.inert = np.zeros(0, np.int64)
        if imu is not None and len(imu["cam_i"]):
            seen[imu["cam_i"]] = True
            seen[imu["cam_j"]] = True
            self.inert = np.unique(np.concatenate([imu["cam_i"], imu["cam_j"]])).astype(np.int64)
            self.vel0 = np.array(imu["cam_velocity"], np.float64)
            self.bias0 = np.array(imu["cam_bias"], np.float64)
            self.factors = [{k: np.asarray(imu[k][f], np.float64) for k in ("rotation", "velocity", "position", "covariance",
                                                                           "bias_gyro", "bias_accel", "bias_jacobian")} |
                            dict(duration=float(imu["duration"][f]), i=int(imu["cam_i"][f]), j=int(imu["cam_j"][f]))
                            for f in range(len(imu["cam_i"]))]
        if delta is not None:                       # this frame's velocity is a free block; bias is not in the problem
            self.inert = np.array([0], np.int64)
            self.vel0 = np.array([delta["velocity"]], np.float64)
            self.bias0 = np.zeros((1, 6))
            f = {k: np.asarray(delta["imu"][k][0], np.float64) for k in ("rotation", "velocity", "position", "covariance",
                                                                         "bias_gyro", "bias_accel", "bias_jacobian")}
            self.delta_factor = f | dict(duration=float(delta["imu"]["duration"][0]))
        self.active = np.flatnonzero(np.asarray(cam_free, bool) & seen)
        self.col_of_cam = -np.ones(len(cams), np.int64)
        self.col_of_cam[self.active] = 6 * np.arange(len(self.active))
        self.npose = 6 * len(self.active)
        self.col_of_inert = -np.ones(len(cams), np.int64)
        self.col_of_inert[self.inert] = self.npose + 9 * np.arange(len(self.inert))
        self.nv = 3 if delta is not None else 9     # unknowns per inertial frame
        self.nc = self.npose + self.nv * len(self.inert)
        self.n = self.nc + (0 if points_constant else 3 * len(pts))

    def pack(self, cams, pts, vel=None, bias=None):
        parts = [cams[self.active].ravel()]
        if len(self.inert):
            vel = self.vel0 if vel is None else vel
            bias = self.bias0 if bias is None else bias
            parts.append((vel[self.inert] if self.delta is not None else
                          np.concatenate([vel[self.inert], bias[self.inert]], axis=1)).ravel())
        return np.concatenate(parts + ([] if self.points_constant else [pts.ravel()]))

    def unpack(self, x):
        cams = self.cams0.astype(x.dtype)
        cams[self.active] = x[:self.npose].reshape(-1, 6)
        return cams, (self.pts0.astype(x.dtype) if self.points_constant else x[self.nc:].reshape(-1, 3))

    def unpack_inertial(self, x):
        vel, bias = self.vel0.astype(x.dtype), self.bias0.astype(x.dtype)
        vb = x[self.npose:self.nc].reshape(-1, self.nv)
        vel[self.inert] = vb[:, :3]
        if self.nv == 9:
            bias[self.inert] = vb[:, 3:]
        return vel, bias

    def extra_residuals(self, x):
        """Residuals of the inertial blocks (no loss function), stacked: 9 + 6 per factor pair."""
        if self.prior is not None:
            cams, _ = self.unpack(x)
            P = np.asarray(self.prior[0], np.float64)
            return matrix_to_aa(P.T @ aa_to_matrix(cams[0, :3])) / self.prior[1]
        if self.delta is not None:
            cams, _ = self.unpack(x)
            vel, _ = self.unpack_inertial(x)
            d = self.delta
            return imu_preintegration(self.delta_factor, np.asarray(d["imu"]["gravity"], np.float64),
                                      np.asarray(d["prev_pose"], np.float64), np.asarray(d["prev_velocity"], np.float64),
                                      np.asarray(d["prev_bias"], np.float64), cams[0], vel[0])
        if not len(self.inert):
            return np.zeros(0, x.dtype)
        cams, _ = self.unpack(x)
        vel, bias = self.unpack_inertial(x)
        g = np.asarray(self.imu["gravity"], np.float64)
        out = []
        for f in self.factors:
            i, j = f["i"], f["j"]
            out.append(imu_preintegration(f, g, cams[i], vel[i], bias[i], cams[j], vel[j]))
            el = np.sqrt(max(f["duration"], 1e-9))
            sg, sa = self.imu["gyro_bias_sigma"] * el, self.imu["accel_bias_sigma"] * el
            out.append(np.concatenate([(bias[j, :3] - bias[i, :3]) / sg, (bias[j, 3:] - bias[i, 3:]) / sa]))
        return np.concatenate(out)

    def has_extras(self):
        return len(self.inert) > 0 or self.prior is not None

    def extra_jacobian(self, x, h=1e-30):
        r0 = np.real(self.extra_residuals(x.astype(np.complex128)))
        J = np.zeros((len(r0), self.n))
        for k in range(self.nc):                 # the inertial blocks only depend on camera-side unknowns
            xz = x.astype(np.complex128)
            xz[k] += 1j * h
            J[:, k] = np.imag(self.extra_residuals(xz)) / h
        return r0, J

    def cost(self, x):
        cams, pts = self.unpack(x)
        r = residuals(cams, pts, self.obs_cam, self.obs_pt, self.obs_uv, self.K)
        rho, _ = huber(np.sum(r * r, axis=1), self.a)
        e = np.real(self.extra_residuals(x.astype(np.complex128))) if self.has_extras() else np.zeros(0)
        return 0.5 * float(np.sum(rho)) + 0.5 * float(e @ e)

    def linearize(self, x):
        """Corrected residual vector r [2M], dense corrected Jacobian J [2M, n], cost."""
        cams, pts = self.unpack(x)
        r = residuals(cams, pts, self.obs_cam, self.obs_pt, self.obs_uv, self.K)
        jc, jp = jacobian_blocks(cams, pts, self.obs_cam, self.obs_pt, self.obs_uv, self.K)
        rho, rho1 = huber(np.sum(r * r, axis=1), self.a)
        sr = np.sqrt(rho1)
        M = len(r)
        J = np.zeros((2 * M, self.n))
        for o in range(M):
            c0 = self.col_of_cam[self.obs_cam[o]]
            if c0 >= 0:
                J[2 * o:2 * o + 2, c0:c0 + 6] = jc[o] * sr[o]
            if not self.points_constant:
                p0 = self.nc + 3 * self.obs_pt[o]
                J[2 * o:2 * o + 2, p0:p0 + 3] = jp[o] * sr[o]
        rv, cost = (r * sr[:, None]).ravel(), 0.5 * float(np.sum(rho))
        if self.has_extras():
            re, Je = self.extra_jacobian(x)
            rv, J, cost = np.concatenate([rv, re]), np.vstack([J, Je]), cost + 0.5 * float(re @ re)
        return rv, J, cost


def solve(prob, max_iter=10, r0=1e4, rmax=1e16, rmin=1e-32, min_rel=1e-3, dmin=1e-6, dmax=1e32,
          ftol=1e-6, gtol=1e-10, ptol=1e-8, max_invalid=5):
    """Returns (x, summary dict, trace list) with the fields of orc_ba_iteration."""
    x = prob.pack(prob.cams0, prob.pts0)
    r, J, x_cost = prob.linearize(x)
    summary = dict(initial_cost=x_cost, iterations=0, successful_steps=0, termination=0)
    trace = []
    scale = 1.0 / (1.0 + np.sqrt(np.sum(J * J, axis=0)))
    radius, factor, invalid = r0, 2.0, 0
    best_x, best_cost = x.copy(), x_cost
    if not np.isfinite(x_cost):
        summary["termination"] = 5
    elif np.max(np.abs(J.T @ r)) <= gtol:
        summary["termination"] = 3
    else:
        while True:
            if summary["iterations"] >= max_iter:
                summary["termination"] = 0
                break
            summary["iterations"] += 1
            Js = J * scale
            d2 = np.clip(np.sum(Js * Js, axis=0), dmin, dmax) / radius
            H = Js.T @ Js + np.diag(d2)
            ok = True
            try:
                L = np.linalg.cholesky(H)
                y = np.linalg.solve(L.T, np.linalg.solve(L, Js.T @ r))
            except np.linalg.LinAlgError:
                ok = False
            mcc = 0.0
            if ok:
                step = -y
                ok = bool(np.all(np.isfinite(step)))
            if ok:
                Jd = Js @ step
                mcc = -float(Jd @ (r + Jd / 2.0))
            if not ok or not (mcc > 0.0):
                trace.append(dict(cost=x_cost, candidate_cost=0.0, model_cost_change=mcc if ok else 0.0, radius=radius,
                                  step_norm=0.0, x_norm=0.0, outcome=-1))
                invalid += 1
                if invalid >= max_invalid:
                    summary["termination"] = 5
                    break
                radius /= factor
                factor *= 2.0
                continue
            invalid = 0
            delta = step * scale
            cand = x + delta
            cand_cost = prob.cost(cand)
            step_norm, x_norm = float(np.linalg.norm(x - cand)), float(np.linalg.norm(x))
            ent = dict(cost=x_cost, candidate_cost=cand_cost, model_cost_change=mcc, radius=radius, step_norm=step_norm,
                       x_norm=x_norm, outcome=0)
            trace.append(ent)
            if step_norm <= ptol * (x_norm + ptol):
                ent["outcome"] = 2
                summary["termination"] = 2
                break
            if abs(x_cost - cand_cost) <= ftol * x_cost:
                ent["outcome"] = 2
                summary["termination"] = 1
                break
            rel = (x_cost - cand_cost) / mcc
            if rel > min_rel and np.isfinite(cand_cost):
                ent["outcome"] = 1
                x = cand
                r, J, x_cost = prob.linearize(x)
                summary["successful_steps"] += 1
                radius = min(rmax, radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3))
                factor = 2.0
                if x_cost < best_cost:
                    best_cost, best_x = x_cost, x.copy()
                if np.max(np.abs(J.T @ r)) <= gtol:
                    summary["termination"] = 3
                    break
            else:
                radius /= factor
                factor *= 2.0
                if radius < rmin:
                    summary["termination"] = 4
                    break
    summary["final_cost"] = best_cost
    summary["final_radius"] = radius
    summary["usable"] = int(summary["termination"] != 5 and np.isfinite(best_cost) and best_cost <= summary["initial_cost"])
    return best_x, summary, trace


class PoseGraphProblem:
    """optimization::pose_graph (reference src/Optimization.cpp:376-639) for `solve`: residuals RelativePoseError /
    RelativePose4DoFError by complex step, sequential edges measured as T_i T_{i+1}^-1 (numpy f32 inverse, widened; or
    handed in as `seq_relative`, so that a comparison is not limited by two f32 inverses differing in their last bits), loop
    edges with HuberLoss(1.0) through the corrector; the first key frame is constant.  x0 [n][6] = pack_pose of the f32
    poses (passed in, so that this file needs no f32 rotation conversions of its own) or [n][4] = (0, centre)."""

    def __init__(self, poses_f32, x0, loops, four_dof=False, up=(0.0, 0.0, 1.0), seq_relative=None):
        P = np.asarray(poses_f32, np.float32).reshape(-1, 4, 4)
        self.n = len(P)
        self.bs = 4 if four_dof else 6
        self.four_dof = bool(four_dof)
        self.up = np.asarray(up, np.float64)
        self.R0 = P[:, :3, :3].astype(np.float64)
        self.x0 = np.asarray(x0, np.float64).reshape(self.n, self.bs)
        self.edges = []
        for i in range(self.n - 1):
            rel = (P[i].astype(np.float64) @ np.linalg.inv(P[i + 1]).astype(np.float64) if seq_relative is None
                   else np.asarray(seq_relative[i], np.float64).reshape(4, 4))
            self.edges.append((i, i + 1, rel[:3, :3], rel[:3, 3], 0.02, 0.2, False))
        for a, b, rel in loops:
            if a == b or not (0 <= a < self.n and 0 <= b < self.n):
                continue
            rel = np.asarray(rel, np.float64).reshape(4, 4)
            self.edges.append((int(a), int(b), rel[:3, :3], rel[:3, 3], 0.05, 0.5, True))
        self.cams0, self.pts0 = None, None

    def pack(self, *_):
        return self.x0[1:].ravel().copy()

    def full(self, x):
        return np.concatenate([self.x0[0].astype(x.dtype), x]).reshape(self.n, self.bs)

    def rot(self, xb, i):
        if not self.four_dof:
            return aa_to_matrix(xb[:3])
        return self.R0[i] @ aa_to_matrix(-self.up * xb[0])

    def edge_residual(self, e, X):
        a, b, Rm, tm, sr, st, _ = e
        Rf, Rt = self.rot(X[a], a), self.rot(X[b], b)
        cf, ct = X[a][-3:], X[b][-3:]
        rv = matrix_to_aa(Rm.T @ (Rf @ Rt.T))
        return np.concatenate([rv / sr, (Rf @ (ct - cf) - tm) / st])

    def blocks(self, x):
        X = self.full(x.astype(np.complex128) if np.iscomplexobj(x) else x)
        return [self.edge_residual(e, X) for e in self.edges]

    def cost(self, x):
        c = 0.0
        for e, r in zip(self.edges, self.blocks(x)):
            s = float(np.real(r) @ np.real(r))
            c += 0.5 * (huber(np.array(s), 1.0)[0] if e[6] else s)
        return float(c)

    def linearize(self, x, h=1e-30):
        nb, bs = len(self.edges), self.bs
        r0 = [np.real(r) for r in self.blocks(x)]
        J = np.zeros((6 * nb, len(x)))
        for k, e in enumerate(self.edges):
            for kf in (e[0], e[1]):
                if kf == 0:
                    continue
                for q in range(bs):
                    col = bs * (kf - 1) + q
                    xz = x.astype(np.complex128)
                    xz[col] += 1j * h
                    J[6 * k:6 * k + 6, col] = np.imag(self.edge_residual(e, self.full(xz))) / h
        cost = 0.0
        rv = np.zeros(6 * nb)
        for k, (e, r) in enumerate(zip(self.edges, r0)):
            s = float(r @ r)
            rho, rho1 = (huber(np.array(s), 1.0) if e[6] else (s, 1.0))
            cost += 0.5 * float(rho)
            sc = np.sqrt(float(rho1))
            rv[6 * k:6 * k + 6] = r * sc
            J[6 * k:6 * k + 6] *= sc
        return rv, J, cost
