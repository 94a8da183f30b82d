// imu_dual.h — the inertial residual blocks of the reference's optimisation on the device:
//   PreintegrationError   src/ImuFactor.cpp:19-87   9 residuals; blocks pose_i 6, velocity_i 3, bias_i 6, pose_j 6, velocity_j 3
//   BiasRandomWalk        src/ImuFactor.cpp:89-118  6 residuals; blocks bias_i 6, bias_j 6
//   PredictedRotationError src/Optimization.cpp:74-95  3 residuals; block pose 6
// The reference differentiates them with ceres::AutoDiffCostFunction; here the same functors run on forward-mode dual
// numbers (value + 24 partials, the local parameter order above).  There are at most a few dozen factors per solve
// (one pair per consecutive pair of optimised key frames), so one thread per factor is plenty and register pressure
// does not matter: these are not hot kernels, they ride along with the reduced-system assembly.
#pragma once
#include <math.h>

#include "../../include/rsgpu.h"

#define IMU_NP 24

// Three carriers for the same functor code (what ceres::Jet<double, N> is upstream):
//   Dual      value + all 24 partials in one thread            (a single factor inside a one-workgroup solver)
//   DualLane  value + ONE partial per lane: lane (threadIdx.x & 31) of a 32-lane group holds partial number lane —
//             every dual operation is two or three scalar instructions and no arrays exist (bundle adjustment: one
//             group per IMU factor pair; a thread-serial Dual spilled its 25-double temporaries to scratch)
//   DualV     value only                                         (candidate costs)
// Branches inside the functors depend on values only, which all lanes of a group share.
struct Dual {
    double a;
    double v[IMU_NP];
    static __device__ inline Dual constant(double x) { Dual r; r.a = x; for (int i = 0; i < IMU_NP; i++) r.v[i] = 0.0; return r; }
    static __device__ inline Dual variable(double x, int k) { Dual r = constant(x); if (k >= 0) r.v[k] = 1.0; return r; }
};
__device__ inline Dual operator+(const Dual& f, const Dual& g) { Dual r; r.a = f.a + g.a; for (int i = 0; i < IMU_NP; i++) r.v[i] = f.v[i] + g.v[i]; return r; }
__device__ inline Dual operator-(const Dual& f, const Dual& g) { Dual r; r.a = f.a - g.a; for (int i = 0; i < IMU_NP; i++) r.v[i] = f.v[i] - g.v[i]; return r; }
__device__ inline Dual operator-(const Dual& f) { Dual r; r.a = -f.a; for (int i = 0; i < IMU_NP; i++) r.v[i] = -f.v[i]; return r; }
__device__ inline Dual operator*(const Dual& f, const Dual& g) { Dual r; r.a = f.a * g.a; for (int i = 0; i < IMU_NP; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
__device__ inline Dual operator*(const Dual& f, double s) { Dual r; r.a = f.a * s; for (int i = 0; i < IMU_NP; i++) r.v[i] = f.v[i] * s; return r; }
__device__ inline Dual operator/(const Dual& f, const Dual& g)
{
    Dual r; const double gi = 1.0 / g.a, fg = f.a * gi;
    r.a = fg; for (int i = 0; i < IMU_NP; i++) r.v[i] = (f.v[i] - fg * g.v[i]) * gi; return r;
}
__device__ inline Dual dsqrt(const Dual& f) { Dual r; const double t = sqrt(f.a), h = 1.0 / (2.0 * t); r.a = t; for (int i = 0; i < IMU_NP; i++) r.v[i] = f.v[i] * h; return r; }
__device__ inline Dual dcos(const Dual& f) { Dual r; const double s = -sin(f.a); r.a = cos(f.a); for (int i = 0; i < IMU_NP; i++) r.v[i] = s * f.v[i]; return r; }
__device__ inline Dual dsin(const Dual& f) { Dual r; const double c = cos(f.a); r.a = sin(f.a); for (int i = 0; i < IMU_NP; i++) r.v[i] = c * f.v[i]; return r; }
__device__ inline Dual datan2(const Dual& g, const Dual& f)
{
    Dual r; const double t = 1.0 / (f.a * f.a + g.a * g.a);
    r.a = atan2(g.a, f.a); for (int i = 0; i < IMU_NP; i++) r.v[i] = t * (f.a * g.v[i] - g.a * f.v[i]); return r;
}

struct DualLane {
    double a, v;
    static __device__ inline DualLane constant(double x) { DualLane r; r.a = x; r.v = 0.0; return r; }
    static __device__ inline DualLane variable(double x, int k) { DualLane r; r.a = x; r.v = ((int)(threadIdx.x & 31) == k) ? 1.0 : 0.0; return r; }
};
__device__ inline DualLane operator+(const DualLane& f, const DualLane& g) { return DualLane{f.a + g.a, f.v + g.v}; }
__device__ inline DualLane operator-(const DualLane& f, const DualLane& g) { return DualLane{f.a - g.a, f.v - g.v}; }
__device__ inline DualLane operator-(const DualLane& f) { return DualLane{-f.a, -f.v}; }
__device__ inline DualLane operator*(const DualLane& f, const DualLane& g) { return DualLane{f.a * g.a, f.a * g.v + f.v * g.a}; }
__device__ inline DualLane operator*(const DualLane& f, double s) { return DualLane{f.a * s, f.v * s}; }
__device__ inline DualLane operator/(const DualLane& f, const DualLane& g) { const double gi = 1.0 / g.a, fg = f.a * gi; return DualLane{fg, (f.v - fg * g.v) * gi}; }
__device__ inline DualLane dsqrt(const DualLane& f) { const double t = sqrt(f.a), h = 1.0 / (2.0 * t); return DualLane{t, f.v * h}; }
__device__ inline DualLane dcos(const DualLane& f) { const double s = -sin(f.a); return DualLane{cos(f.a), s * f.v}; }
__device__ inline DualLane dsin(const DualLane& f) { const double c = cos(f.a); return DualLane{sin(f.a), c * f.v}; }
__device__ inline DualLane datan2(const DualLane& g, const DualLane& f) { const double t = 1.0 / (f.a * f.a + g.a * g.a); return DualLane{atan2(g.a, f.a), t * (f.a * g.v - g.a * f.v)}; }

struct DualV {
    double a;
    static __device__ inline DualV constant(double x) { return DualV{x}; }
    static __device__ inline DualV variable(double x, int) { return DualV{x}; }
};
__device__ inline DualV operator+(const DualV& f, const DualV& g) { return DualV{f.a + g.a}; }
__device__ inline DualV operator-(const DualV& f, const DualV& g) { return DualV{f.a - g.a}; }
__device__ inline DualV operator-(const DualV& f) { return DualV{-f.a}; }
__device__ inline DualV operator*(const DualV& f, const DualV& g) { return DualV{f.a * g.a}; }
__device__ inline DualV operator*(const DualV& f, double s) { return DualV{f.a * s}; }
__device__ inline DualV operator/(const DualV& f, const DualV& g) { const double gi = 1.0 / g.a; return DualV{f.a * gi}; }
__device__ inline DualV dsqrt(const DualV& f) { return DualV{sqrt(f.a)}; }
__device__ inline DualV dcos(const DualV& f) { return DualV{cos(f.a)}; }
__device__ inline DualV dsin(const DualV& f) { return DualV{sin(f.a)}; }
__device__ inline DualV datan2(const DualV& g, const DualV& f) { return DualV{atan2(g.a, f.a)}; }

#define DRM(R, r, c) (R)[(c) * 3 + (r)]      // column-major 3x3, as ceres / Eigen store it

// ceres::AngleAxisToRotationMatrix
template <class D>
__device__ inline void d_aa_to_matrix(const D aa[3], D R[9])
{
    const D th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (th2.a > 2.220446049250313e-16) {
        const D th = dsqrt(th2);
        const D wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th;
        const D ct = dcos(th), st = dsin(th), omc = D::constant(1.0) - ct;
        R[0] = ct + wx * wx * omc;
        R[1] = wz * st + wx * wy * omc;
        R[2] = -(wy * st) + wx * wz * omc;
        R[3] = wx * wy * omc - wz * st;
        R[4] = ct + wy * wy * omc;
        R[5] = wx * st + wy * wz * omc;
        R[6] = wy * st + wx * wz * omc;
        R[7] = -(wx * st) + wy * wz * omc;
        R[8] = ct + wz * wz * omc;
    } else {
        R[0] = D::constant(1.0); R[1] = aa[2]; R[2] = -aa[1];
        R[3] = -aa[2]; R[4] = D::constant(1.0); R[5] = aa[0];
        R[6] = aa[1]; R[7] = -aa[0]; R[8] = D::constant(1.0);
    }
}

// ceres::RotationMatrixToAngleAxis = RotationMatrixToQuaternion + QuaternionToAngleAxis
template <class D>
__device__ inline void d_matrix_to_aa(const D R[9], D aa[3])
{
    D q[4];
    const D trace = DRM(R, 0, 0) + DRM(R, 1, 1) + DRM(R, 2, 2);
    if (trace.a >= 0.0) {
        D t = dsqrt(trace + D::constant(1.0));
        q[0] = t * 0.5;
        t = D::constant(0.5) / t;
        q[1] = (DRM(R, 2, 1) - DRM(R, 1, 2)) * t;
        q[2] = (DRM(R, 0, 2) - DRM(R, 2, 0)) * t;
        q[3] = (DRM(R, 1, 0) - DRM(R, 0, 1)) * t;
    } else {
        // i = index of the largest diagonal entry, (j, k) = the cyclic successors; written out per case so that R and q
        // stay in registers (a run-time index sends both arrays to scratch memory)
        int i = 0;
        if (DRM(R, 1, 1).a > DRM(R, 0, 0).a) i = 1;
        if ((i == 0 ? DRM(R, 0, 0).a : DRM(R, 1, 1).a) < DRM(R, 2, 2).a) i = 2;
        if (i == 0) {
            D t = dsqrt(DRM(R, 0, 0) - DRM(R, 1, 1) - DRM(R, 2, 2) + D::constant(1.0));
            q[1] = t * 0.5;
            t = D::constant(0.5) / t;
            q[0] = (DRM(R, 2, 1) - DRM(R, 1, 2)) * t;
            q[2] = (DRM(R, 1, 0) + DRM(R, 0, 1)) * t;
            q[3] = (DRM(R, 2, 0) + DRM(R, 0, 2)) * t;
        } else if (i == 1) {
            D t = dsqrt(DRM(R, 1, 1) - DRM(R, 2, 2) - DRM(R, 0, 0) + D::constant(1.0));
            q[2] = t * 0.5;
            t = D::constant(0.5) / t;
            q[0] = (DRM(R, 0, 2) - DRM(R, 2, 0)) * t;
            q[3] = (DRM(R, 2, 1) + DRM(R, 1, 2)) * t;
            q[1] = (DRM(R, 0, 1) + DRM(R, 1, 0)) * t;
        } else {
            D t = dsqrt(DRM(R, 2, 2) - DRM(R, 0, 0) - DRM(R, 1, 1) + D::constant(1.0));
            q[3] = t * 0.5;
            t = D::constant(0.5) / t;
            q[0] = (DRM(R, 1, 0) - DRM(R, 0, 1)) * t;
            q[1] = (DRM(R, 0, 2) + DRM(R, 2, 0)) * t;
            q[2] = (DRM(R, 1, 2) + DRM(R, 2, 1)) * t;
        }
    }
    const D s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2.a > 0.0) {
        const D s = dsqrt(s2);
        const D two_theta = ((q[0].a < 0.0) ? datan2(-s, -q[0]) : datan2(s, q[0])) * 2.0;
        const D k = two_theta / s;
        for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * k;
    } else {
        for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * 2.0;
    }
}

template <class D>
__device__ inline void d_mm(const D* A, const D* B, D* C, bool ta, bool tb)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            D s = (ta ? DRM(A, 0, r) : DRM(A, r, 0)) * (tb ? DRM(B, c, 0) : DRM(B, 0, c));
            for (int k = 1; k < 3; k++) s = s + (ta ? DRM(A, k, r) : DRM(A, r, k)) * (tb ? DRM(B, c, k) : DRM(B, k, c));
            DRM(C, r, c) = s;
        }
}
template <class D>
__device__ inline void d_mv(const D* A, const D v[3], D out[3])
{
    for (int r = 0; r < 3; r++) out[r] = DRM(A, r, 0) * v[0] + DRM(A, r, 1) * v[1] + DRM(A, r, 2) * v[2];
}

// device copy of one factor pair: rs_imu_factor + the whitener L^-1 (computed on the host, src/ImuFactor.cpp:10-17)
struct ImuFactorDev {
    rs_imu_factor f;
    double W[81];
};

// PreintegrationError::operator() (reference src/ImuFactor.cpp:27-75) on the carrier D: the UNWHITENED residual res[9].
// Local parameter order = partial index: pose_i 0-5, velocity_i 6-8, bias_i 9-14, pose_j 15-20, velocity_j 21-23.
template <class D>
__device__ inline void imu_preintegration_raw(const rs_imu_factor& f, const double g[3], const double* pose_i, const double* vel_i,
                                              const double* bias_i, const double* pose_j, const double* vel_j, D res[9])
{
    D pi[6], vi[3], bi[6], pj[6], vj[3];
    for (int k = 0; k < 6; k++) { pi[k] = D::variable(pose_i[k], k); bi[k] = D::variable(bias_i[k], 9 + k); pj[k] = D::variable(pose_j[k], 15 + k); }
    for (int k = 0; k < 3; k++) { vi[k] = D::variable(vel_i[k], 6 + k); vj[k] = D::variable(vel_j[k], 21 + k); }
    D Ri[9], Rj[9];
    d_aa_to_matrix(pi, Ri);
    d_aa_to_matrix(pj, Rj);
    D db[6], corr[9];
    for (int k = 0; k < 3; k++) { db[k] = bi[k] - D::constant(f.bias_gyro[k]); db[k + 3] = bi[k + 3] - D::constant(f.bias_accel[k]); }
    for (int a = 0; a < 9; a++) {
        D s = db[0] * f.bias_jacobian[a * 6];
        for (int k = 1; k < 6; k++) s = s + db[k] * f.bias_jacobian[a * 6 + k];
        corr[a] = s;
    }
    D Rc[9], dR[9], Rm[9], Rs[9], Re[9];
    d_aa_to_matrix(corr, Rc);
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) DRM(dR, rr, c) = D::constant(f.rotation[rr * 3 + c]);
    d_mm(dR, Rc, Rm, false, false);
    const double T = f.duration;
    d_mm(Ri, Rj, Rs, false, true);
    D dv[3], dp[3], sv[3], sp[3];
    for (int k = 0; k < 3; k++) {
        dv[k] = vj[k] - vi[k] - D::constant(g[k] * T);
        dp[k] = pj[3 + k] - pi[3 + k] - vi[k] * T - D::constant(0.5 * g[k] * T * T);
    }
    d_mv(Ri, dv, sv);
    d_mv(Ri, dp, sp);
    d_mm(Rm, Rs, Re, true, false);
    d_matrix_to_aa(Re, res);
    for (int k = 0; k < 3; k++) {
        res[3 + k] = sv[k] - (D::constant(f.velocity[k]) + corr[3 + k]);
        res[6 + k] = sp[k] - (D::constant(f.position[k]) + corr[6 + k]);
    }
}

// whitened preintegration residual r[9] and, when J != nullptr, its Jacobian J[9][24] — one thread does everything
// (values only when J is null)
__device__ inline void imu_preintegration(const ImuFactorDev& F, const double g[3], const double* pose_i, const double* vel_i,
                                          const double* bias_i, const double* pose_j, const double* vel_j, double r[9], double* J)
{
    if (!J) {
        DualV res[9];
        imu_preintegration_raw<DualV>(F.f, g, pose_i, vel_i, bias_i, pose_j, vel_j, res);
        for (int a = 0; a < 9; a++) {
            double s = 0.0;
            for (int k = 0; k < 9; k++) s += F.W[a * 9 + k] * res[k].a;
            r[a] = s;
        }
        return;
    }
    Dual res[9];
    imu_preintegration_raw<Dual>(F.f, g, pose_i, vel_i, bias_i, pose_j, vel_j, res);
    for (int a = 0; a < 9; a++) {
        double s = 0.0;
        for (int k = 0; k < 9; k++) s += F.W[a * 9 + k] * res[k].a;
        r[a] = s;
        for (int q = 0; q < IMU_NP; q++) {
            double t = 0.0;
            for (int k = 0; k < 9; k++) t += F.W[a * 9 + k] * res[k].v[q];
            J[a * IMU_NP + q] = t;
        }
    }
}

// The same with the 24 partials spread over the lanes of a 32-lane group (all 32 lanes must call): r[9] whitened (equal
// in every lane), jl[9] = this lane's column of the whitened Jacobian (lanes 24..31: zero).
__device__ inline void imu_preintegration_lanes(const ImuFactorDev& F, const double g[3], const double* pose_i, const double* vel_i,
                                                const double* bias_i, const double* pose_j, const double* vel_j, double r[9], double jl[9])
{
    DualLane res[9];
    imu_preintegration_raw<DualLane>(F.f, g, pose_i, vel_i, bias_i, pose_j, vel_j, res);
    for (int a = 0; a < 9; a++) {
        double s = 0.0, t = 0.0;
        for (int k = 0; k < 9; k++) { s += F.W[a * 9 + k] * res[k].a; t += F.W[a * 9 + k] * res[k].v; }
        r[a] = s;
        jl[a] = t;
    }
}

// bias random walk: r[6]; the Jacobian is -1/sigma on bias_i, +1/sigma on bias_j
__device__ inline void imu_bias_walk(const rs_imu_factor& f, const double* bias_i, const double* bias_j, double r[6], double inv_sigma[2])
{
    const double elapsed = sqrt(fmax(f.duration, 1e-9));
    inv_sigma[0] = 1.0 / (f.gyro_bias_sigma * elapsed);
    inv_sigma[1] = 1.0 / (f.accel_bias_sigma * elapsed);
    for (int i = 0; i < 3; i++) {
        r[i] = (bias_j[i] - bias_i[i]) * inv_sigma[0];
        r[i + 3] = (bias_j[i + 3] - bias_i[i + 3]) * inv_sigma[1];
    }
}

// rotation prior: r[3], J[3][6] (columns 3..5 are zero); predicted row-major
__device__ inline void imu_rotation_prior(const double predicted[9], double sigma, const double* pose, double r[3], double* J)
{
    Dual p[3], R[9], P[9], Dm[9], off[3];
    for (int k = 0; k < 3; k++) p[k] = Dual::variable(pose[k], k);
    d_aa_to_matrix(p, R);
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) DRM(P, rr, c) = Dual::constant(predicted[rr * 3 + c]);
    d_mm(P, R, Dm, true, false);
    d_matrix_to_aa(Dm, off);
    for (int k = 0; k < 3; k++) {
        r[k] = off[k].a / sigma;
        if (J) for (int q = 0; q < 6; q++) J[k * 6 + q] = (q < 3 ? off[k].v[q] : 0.0) / sigma;
    }
}

// rotation prior with the three partials on lanes 0..2 of a 32-lane group (all 32 lanes must call): r[3] (equal in every
// lane), jl[3] = this lane's column of the 3 x 3 Jacobian w.r.t. the angle-axis part (lanes >= 3: zero)
__device__ inline void imu_rotation_prior_lanes(const double predicted[9], double sigma, const double* pose, double r[3], double jl[3])
{
    DualLane p[3], R[9], P[9], Dm[9], off[3];
    for (int k = 0; k < 3; k++) p[k] = DualLane::variable(pose[k], k);
    d_aa_to_matrix(p, R);
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) DRM(P, rr, c) = DualLane::constant(predicted[rr * 3 + c]);
    d_mm(P, R, Dm, true, false);
    d_matrix_to_aa(Dm, off);
    for (int k = 0; k < 3; k++) { r[k] = off[k].a / sigma; jl[k] = off[k].v / sigma; }
}
