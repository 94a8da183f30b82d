/*
 * oracle/imu.c — CPU restatement of the inertial residual blocks of the reference's optimisation:
 *   PreintegrationError + whitener      src/ImuFactor.cpp:10-87     (9 residuals; blocks pose_i 6, velocity_i 3,
 *                                                                    bias_i 6, pose_j 6, velocity_j 3)
 *   BiasRandomWalk                      src/ImuFactor.cpp:89-118    (6 residuals; blocks bias_i 6, bias_j 6)
 *   PredictedRotationError              src/Optimization.cpp:74-95  (3 residuals; block pose 6)
 * as they enter refine_pose (:237-258) and bundle_adjust (:317-346).  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (see rs_oracle.h): the reference holds no fixtures for these factors.
 *
 * Third-party semantics restated (Ceres 2.x, not in /root/reference): AutoDiffCostFunction = forward-mode jets
 * through the functor (here 24-wide: 6+3+6+6+3); ceres/rotation.h AngleAxisToRotationMatrix (column-major, first-order
 * branch below DBL_EPSILON), RotationMatrixToAngleAxis = RotationMatrixToQuaternion + QuaternionToAngleAxis (atan2
 * form, k = 2 branch at zero angle); Eigen LLT for the whitener (identity when the factorisation fails).
 * Matrices cross this interface ROW-major; inside, as in Ceres / Eigen, rotation arrays are column-major.
 */
#include <math.h>
#include <string.h>

#include "rs_oracle.h"

#define NJ 24
#include "jetn.h"

/* whitener, src/ImuFactor.cpp:10-17: L^-1 of the LLT of the covariance, identity when not positive definite.
 * cov, W row-major 9x9. */
void orc_imu_whitener(const double cov[81], double W[81])
{
    double L[81];
    memcpy(L, cov, sizeof L);
    int ok = 1;
    for (int j = 0; j < 9 && ok; j++) {
        double d = L[j * 9 + j];
        for (int k = 0; k < j; k++) d -= L[j * 9 + k] * L[j * 9 + k];
        if (!(d > 0.0) || !isfinite(d)) { ok = 0; break; }
        d = sqrt(d);
        L[j * 9 + j] = d;
        for (int i = j + 1; i < 9; i++) {
            double s = L[i * 9 + j];
            for (int k = 0; k < j; k++) s -= L[i * 9 + k] * L[j * 9 + k];
            L[i * 9 + j] = s / d;
        }
    }
    memset(W, 0, sizeof(double) * 81);
    if (!ok) { for (int i = 0; i < 9; i++) W[i * 9 + i] = 1.0; return; }
    for (int c = 0; c < 9; c++)        /* L X = I, column by column */
        for (int i = 0; i < 9; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s -= L[i * 9 + k] * W[k * 9 + c];
            W[i * 9 + c] = s / L[i * 9 + i];
        }
}

/* PreintegrationError::operator()<Jet>, src/ImuFactor.cpp:27-81.  Local parameter order (= jet slots):
 * pose_i 0-5, velocity_i 6-8, bias_i 9-14, pose_j 15-20, velocity_j 21-23.  r[9], J[9][24] row-major. */
void orc_imu_preintegration(const orc_imu_factor* f, const double gravity[3], const double pose_i[6], const double vel_i[3],
                            const double bias_i[6], const double pose_j[6], const double vel_j[3], double r[9], double J[216])
{
    jet pi[6], vi[3], bi[6], pj[6], vj[3];
    for (int k = 0; k < 6; k++) { pi[k] = jv(pose_i[k], k); bi[k] = jv(bias_i[k], 9 + k); pj[k] = jv(pose_j[k], 15 + k); }
    for (int k = 0; k < 3; k++) { vi[k] = jv(vel_i[k], 6 + k); vj[k] = jv(vel_j[k], 21 + k); }
    jet Ri[9], Rj[9];
    aa_to_matrix(pi, Ri);                               /* world_to_camera_i, :45-46 */
    aa_to_matrix(pj, Rj);
    jet db[6], corr[9];
    for (int k = 0; k < 3; k++) { db[k] = jsub(bi[k], jc(f->bias_gyro[k])); db[k + 3] = jsub(bi[k + 3], jc(f->bias_accel[k])); }   /* :49-53 */
    for (int a = 0; a < 9; a++) {                       /* correction = bias_jacobian * bias_change, :54 */
        jet s = jc(0.0);
        for (int k = 0; k < 6; k++) s = (k == 0) ? jscale(db[k], f->bias_jacobian[a * 6 + k]) : jadd(s, jscale(db[k], f->bias_jacobian[a * 6 + k]));
        corr[a] = s;
    }
    jet Rc[9], dR[9], Rm[9];
    aa_to_matrix(corr, Rc);                             /* rotation_correction, :56-57 */
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) RM(dR, rr, c) = jc(f->rotation[rr * 3 + c]);
    mm(dR, Rc, Rm, 0, 0);                               /* measured_rotation, :58 */
    jet mvel[3], mpos[3];
    for (int k = 0; k < 3; k++) { mvel[k] = jadd(jc(f->velocity[k]), corr[3 + k]); mpos[k] = jadd(jc(f->position[k]), corr[6 + k]); }
    const double T = f->duration;
    jet Rs[9];
    mm(Ri, Rj, Rs, 0, 1);                               /* state_rotation = R_i R_j^T, :66 */
    jet dv[3], dp[3], sv[3], sp[3];
    for (int k = 0; k < 3; k++) {
        dv[k] = jsub(jsub(vj[k], vi[k]), jc(gravity[k] * T));                                             /* :67 */
        dp[k] = jsub(jsub(jsub(pj[3 + k], pi[3 + k]), jscale(vi[k], T)), jc(0.5 * gravity[k] * T * T));      /* :68-69 */
    }
    mv(Ri, dv, sv);
    mv(Ri, dp, sp);
    jet Re[9], res[9];
    mm(Rm, Rs, Re, 1, 0);                               /* rotation_error = measured^T state, :72 */
    matrix_to_aa(Re, res);                              /* :73 */
    for (int k = 0; k < 3; k++) { res[3 + k] = jsub(sv[k], mvel[k]); res[6 + k] = jsub(sp[k], mpos[k]); }   /* :74-75 */
    double W[81];
    orc_imu_whitener(f->covariance, W);
    for (int a = 0; a < 9; a++) {                       /* whitened = W residual, :78-79 */
        double s = 0.0;
        double g[NJ];
        memset(g, 0, sizeof g);
        for (int k = 0; k < 9; k++) {
            s += W[a * 9 + k] * res[k].a;
            for (int q = 0; q < NJ; q++) g[q] += W[a * 9 + k] * res[k].v[q];
        }
        r[a] = s;
        if (J) memcpy(J + a * NJ, g, sizeof g);
    }
}

/* BiasRandomWalk, src/ImuFactor.cpp:89-118: local parameters bias_i 0-5, bias_j 6-11.  r[6], J[6][12]. */
void orc_imu_bias_walk(const orc_imu_factor* f, const double bias_i[6], const double bias_j[6], double r[6], double J[72])
{
    const double elapsed = sqrt(fmax(f->duration, 1e-9));                                     /* :113 */
    const double sg = f->gyro_bias_sigma * elapsed, sa = f->accel_bias_sigma * elapsed;
    if (J) memset(J, 0, sizeof(double) * 72);
    for (int i = 0; i < 3; i++) {
        r[i] = (bias_j[i] - bias_i[i]) / sg;
        r[i + 3] = (bias_j[i + 3] - bias_i[i + 3]) / sa;
        if (J) {
            J[i * 12 + i] = -1.0 / sg; J[i * 12 + 6 + i] = 1.0 / sg;
            J[(i + 3) * 12 + i + 3] = -1.0 / sa; J[(i + 3) * 12 + 6 + i + 3] = 1.0 / sa;
        }
    }
}

/* PredictedRotationError, src/Optimization.cpp:75-94: predicted row-major 3x3 (world to camera), pose 6.  r[3], J[3][6]. */
void orc_rotation_prior(const double predicted[9], double sigma, const double pose[6], double r[3], double J[18])
{
    jet p[3], R[9], P[9], D[9], off[3];
    for (int k = 0; k < 3; k++) p[k] = jv(pose[k], k);
    aa_to_matrix(p, R);
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) RM(P, rr, c) = jc(predicted[rr * 3 + c]);
    mm(P, R, D, 1, 0);                                  /* predicted^T * rotation, :84 */
    matrix_to_aa(D, off);
    for (int k = 0; k < 3; k++) {
        r[k] = off[k].a / sigma;
        if (J) for (int q = 0; q < 6; q++) J[k * 6 + q] = (q < 3 ? off[k].v[q] : 0.0) / sigma;
    }
}
