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

struct Dual {
    double a;
    double v[IMU_NP];
};

__device__ inline Dual dconst(double x) { Dual r; r.a = x; for (int i = 0; i < IMU_NP; i++) r.v[i] = 0.0; return r; }
__device__ inline Dual dvar(double x, int k) { Dual r = dconst(x); if (k >= 0) r.v[k] = 1.0; return r; }
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

#define DRM(R, r, c) (R)[(c) * 3 + (r)]      // column-major 3x3, as ceres / Eigen store it

// ceres::AngleAxisToRotationMatrix
__device__ inline void d_aa_to_matrix(const Dual aa[3], Dual R[9])
{
    const Dual th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (th2.a > 2.220446049250313e-16) {
        const Dual th = dsqrt(th2);
        const Dual wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th;
        const Dual ct = dcos(th), st = dsin(th), omc = dconst(1.0) - ct;
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
        R[0] = dconst(1.0); R[1] = aa[2]; R[2] = -aa[1];
        R[3] = -aa[2]; R[4] = dconst(1.0); R[5] = aa[0];
        R[6] = aa[1]; R[7] = -aa[0]; R[8] = dconst(1.0);
    }
}

// ceres::RotationMatrixToAngleAxis = RotationMatrixToQuaternion + QuaternionToAngleAxis
__device__ inline void d_matrix_to_aa(const Dual R[9], Dual aa[3])
{
    Dual q[4];
    const Dual trace = DRM(R, 0, 0) + DRM(R, 1, 1) + DRM(R, 2, 2);
    if (trace.a >= 0.0) {
        Dual t = dsqrt(trace + dconst(1.0));
        q[0] = t * 0.5;
        t = dconst(0.5) / t;
        q[1] = (DRM(R, 2, 1) - DRM(R, 1, 2)) * t;
        q[2] = (DRM(R, 0, 2) - DRM(R, 2, 0)) * t;
        q[3] = (DRM(R, 1, 0) - DRM(R, 0, 1)) * t;
    } else {
        int i = 0;
        if (DRM(R, 1, 1).a > DRM(R, 0, 0).a) i = 1;
        if (DRM(R, 2, 2).a > DRM(R, i, i).a) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        Dual t = dsqrt(DRM(R, i, i) - DRM(R, j, j) - DRM(R, k, k) + dconst(1.0));
        q[i + 1] = t * 0.5;
        t = dconst(0.5) / t;
        q[0] = (DRM(R, k, j) - DRM(R, j, k)) * t;
        q[j + 1] = (DRM(R, j, i) + DRM(R, i, j)) * t;
        q[k + 1] = (DRM(R, k, i) + DRM(R, i, k)) * t;
    }
    const Dual s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2.a > 0.0) {
        const Dual s = dsqrt(s2);
        const Dual two_theta = ((q[0].a < 0.0) ? datan2(-s, -q[0]) : datan2(s, q[0])) * 2.0;
        const Dual k = two_theta / s;
        for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * k;
    } else {
        for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * 2.0;
    }
}

__device__ inline void d_mm(const Dual* A, const Dual* B, Dual* C, bool ta, bool tb)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            Dual s = (ta ? DRM(A, 0, r) : DRM(A, r, 0)) * (tb ? DRM(B, c, 0) : DRM(B, 0, c));
            for (int k = 1; k < 3; k++) s = s + (ta ? DRM(A, k, r) : DRM(A, r, k)) * (tb ? DRM(B, c, k) : DRM(B, k, c));
            DRM(C, r, c) = s;
        }
}
__device__ inline void d_mv(const Dual* A, const Dual v[3], Dual out[3])
{
    for (int r = 0; r < 3; r++) out[r] = DRM(A, r, 0) * v[0] + DRM(A, r, 1) * v[1] + DRM(A, r, 2) * v[2];
}

// device copy of one factor pair: rs_imu_factor + the whitener L^-1 (computed on the host, src/ImuFactor.cpp:10-17)
struct ImuFactorDev {
    rs_imu_factor f;
    double W[81];
};

// whitened preintegration residual r[9] and, when J != nullptr, its Jacobian J[9][24]
__device__ inline void imu_preintegration(const ImuFactorDev& F, const double g[3], const double* pose_i, const double* vel_i,
                                          const double* bias_i, const double* pose_j, const double* vel_j, double r[9], double* J)
{
    const rs_imu_factor& f = F.f;
    Dual pi[6], vi[3], bi[6], pj[6], vj[3];
    for (int k = 0; k < 6; k++) { pi[k] = dvar(pose_i[k], k); bi[k] = dvar(bias_i[k], 9 + k); pj[k] = dvar(pose_j[k], 15 + k); }
    for (int k = 0; k < 3; k++) { vi[k] = dvar(vel_i[k], 6 + k); vj[k] = dvar(vel_j[k], 21 + k); }
    Dual Ri[9], Rj[9];
    d_aa_to_matrix(pi, Ri);
    d_aa_to_matrix(pj, Rj);
    Dual db[6], corr[9];
    for (int k = 0; k < 3; k++) { db[k] = bi[k] - dconst(f.bias_gyro[k]); db[k + 3] = bi[k + 3] - dconst(f.bias_accel[k]); }
    for (int a = 0; a < 9; a++) {
        Dual s = db[0] * f.bias_jacobian[a * 6];
        for (int k = 1; k < 6; k++) s = s + db[k] * f.bias_jacobian[a * 6 + k];
        corr[a] = s;
    }
    Dual Rc[9], dR[9], Rm[9], Rs[9], Re[9], res[9];
    d_aa_to_matrix(corr, Rc);
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) DRM(dR, rr, c) = dconst(f.rotation[rr * 3 + c]);
    d_mm(dR, Rc, Rm, false, false);
    const double T = f.duration;
    d_mm(Ri, Rj, Rs, false, true);
    Dual dv[3], dp[3], sv[3], sp[3];
    for (int k = 0; k < 3; k++) {
        dv[k] = vj[k] - vi[k] - dconst(g[k] * T);
        dp[k] = pj[3 + k] - pi[3 + k] - vi[k] * T - dconst(0.5 * g[k] * T * T);
    }
    d_mv(Ri, dv, sv);
    d_mv(Ri, dp, sp);
    d_mm(Rm, Rs, Re, true, false);
    d_matrix_to_aa(Re, res);
    for (int k = 0; k < 3; k++) {
        res[3 + k] = sv[k] - (dconst(f.velocity[k]) + corr[3 + k]);
        res[6 + k] = sp[k] - (dconst(f.position[k]) + corr[6 + k]);
    }
    for (int a = 0; a < 9; a++) {
        double s = 0.0;
        for (int k = 0; k < 9; k++) s += F.W[a * 9 + k] * res[k].a;
        r[a] = s;
        if (J)
            for (int q = 0; q < IMU_NP; q++) {
                double t = 0.0;
                for (int k = 0; k < 9; k++) t += F.W[a * 9 + k] * res[k].v[q];
                J[a * IMU_NP + q] = t;
            }
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
    Dual p[3], R[9], P[9], D[9], off[3];
    for (int k = 0; k < 3; k++) p[k] = dvar(pose[k], k);
    d_aa_to_matrix(p, R);
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) DRM(P, rr, c) = dconst(predicted[rr * 3 + c]);
    d_mm(P, R, D, true, false);
    d_matrix_to_aa(D, off);
    for (int k = 0; k < 3; k++) {
        r[k] = off[k].a / sigma;
        if (J) for (int q = 0; q < 6; q++) J[k * 6 + q] = (q < 3 ? off[k].v[q] : 0.0) / sigma;
    }
}
