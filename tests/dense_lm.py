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
Small problems only (dense (6C+3P)^2 system).
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


def huber(s, a):
    b = a * a
    out = s > b
    r = np.sqrt(np.where(out, s, 1.0))
    rho = np.where(out, 2.0 * a * r - b, s)
    rho1 = np.where(out, a / r, 1.0)
    return rho, rho1


class Problem:
    def __init__(self, cams, cam_free, pts, obs_ptr, obs_cam, obs_uv, K, huber_a=np.sqrt(5.991)):
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
        self.active = np.flatnonzero(np.asarray(cam_free, bool) & seen)
        self.col_of_cam = -np.ones(len(cams), np.int64)
        self.col_of_cam[self.active] = 6 * np.arange(len(self.active))
        self.nc = 6 * len(self.active)
        self.n = self.nc + 3 * len(pts)

    def pack(self, cams, pts):
        return np.concatenate([cams[self.active].ravel(), pts.ravel()])

    def unpack(self, x):
        cams = self.cams0.copy()
        cams[self.active] = x[:self.nc].reshape(-1, 6)
        return cams, x[self.nc:].reshape(-1, 3)

    def cost(self, x):
        cams, pts = self.unpack(x)
        r = residuals(cams, pts, self.obs_cam, self.obs_pt, self.obs_uv, self.K)
        rho, _ = huber(np.sum(r * r, axis=1), self.a)
        return 0.5 * float(np.sum(rho))

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
            p0 = self.nc + 3 * self.obs_pt[o]
            J[2 * o:2 * o + 2, p0:p0 + 3] = jp[o] * sr[o]
        return (r * sr[:, None]).ravel(), J, 0.5 * float(np.sum(rho))


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
