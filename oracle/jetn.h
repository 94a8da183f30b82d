/*
 * oracle/jetn.h — forward-mode jets of width NJ (define NJ before including) and the ceres/rotation.h conversions
 * restated on them: the autodiff machinery of the oracle's inertial and pose-graph residual blocks
 * (ceres::AutoDiffCostFunction + ceres/jet.h + ceres/rotation.h, Ceres 2.x, not in /root/reference).
 * TEST INFRASTRUCTURE ONLY (see rs_oracle.h).  Everything is static: one copy per translation unit and width.
 */
#ifndef NJ
#error "define NJ (number of partials) before including jetn.h"
#endif
#include <math.h>
#include <string.h>

#define DBL_EPS 2.220446049250313e-16
typedef struct { double a; double v[NJ]; } jet;

static jet jc(double x) { jet r; r.a = x; memset(r.v, 0, sizeof r.v); return r; }
static jet jv(double x, int k) { jet r = jc(x); if (k >= 0) r.v[k] = 1.0; return r; }
static jet jadd(jet f, jet g) { jet r; r.a = f.a + g.a; for (int i = 0; i < NJ; i++) r.v[i] = f.v[i] + g.v[i]; return r; }
static jet jsub(jet f, jet g) { jet r; r.a = f.a - g.a; for (int i = 0; i < NJ; i++) r.v[i] = f.v[i] - g.v[i]; return r; }
static jet jneg(jet f) { jet r; r.a = -f.a; for (int i = 0; i < NJ; i++) r.v[i] = -f.v[i]; return r; }
static jet jmul(jet f, jet g) { jet r; r.a = f.a * g.a; for (int i = 0; i < NJ; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
static jet jscale(jet f, double s) { jet r; r.a = f.a * s; for (int i = 0; i < NJ; i++) r.v[i] = f.v[i] * s; return r; }
static jet jdiv(jet f, jet g)
{
    jet r; const double gi = 1.0 / g.a; const double fg = f.a * gi;
    r.a = fg; for (int i = 0; i < NJ; i++) r.v[i] = (f.v[i] - fg * g.v[i]) * gi; return r;
}
static jet jsqrt(jet f) { jet r; const double t = sqrt(f.a); const double h = 1.0 / (2.0 * t); r.a = t; for (int i = 0; i < NJ; i++) r.v[i] = f.v[i] * h; return r; }
static jet jcos(jet f) { jet r; const double s = -sin(f.a); r.a = cos(f.a); for (int i = 0; i < NJ; i++) r.v[i] = s * f.v[i]; return r; }
static jet jsin(jet f) { jet r; const double c = cos(f.a); r.a = sin(f.a); for (int i = 0; i < NJ; i++) r.v[i] = c * f.v[i]; return r; }
/* atan2(g, f): d = (f dg - g df) / (f^2 + g^2)   (ceres/jet.h) */
static jet jatan2(jet g, jet f)
{
    jet r; const double t = 1.0 / (f.a * f.a + g.a * g.a);
    r.a = atan2(g.a, f.a); for (int i = 0; i < NJ; i++) r.v[i] = t * (f.a * g.v[i] - g.a * f.v[i]); return r;
}

/* ceres::AngleAxisToRotationMatrix, R column-major: R[c * 3 + r] */
static void aa_to_matrix(const jet aa[3], jet R[9])
{
    const jet theta2 = jadd(jadd(jmul(aa[0], aa[0]), jmul(aa[1], aa[1])), jmul(aa[2], aa[2]));
    if (theta2.a > DBL_EPS) {
        const jet theta = jsqrt(theta2);
        const jet wx = jdiv(aa[0], theta), wy = jdiv(aa[1], theta), wz = jdiv(aa[2], theta);
        const jet ct = jcos(theta), st = jsin(theta), omc = jsub(jc(1.0), ct);
        R[0] = jadd(ct, jmul(jmul(wx, wx), omc));
        R[1] = jadd(jmul(wz, st), jmul(jmul(wx, wy), omc));
        R[2] = jadd(jneg(jmul(wy, st)), jmul(jmul(wx, wz), omc));
        R[3] = jsub(jmul(jmul(wx, wy), omc), jmul(wz, st));
        R[4] = jadd(ct, jmul(jmul(wy, wy), omc));
        R[5] = jadd(jmul(wx, st), jmul(jmul(wy, wz), omc));
        R[6] = jadd(jmul(wy, st), jmul(jmul(wx, wz), omc));
        R[7] = jadd(jneg(jmul(wx, st)), jmul(jmul(wy, wz), omc));
        R[8] = jadd(ct, jmul(jmul(wz, wz), omc));
    } else {
        R[0] = jc(1.0); R[1] = aa[2]; R[2] = jneg(aa[1]);
        R[3] = jneg(aa[2]); R[4] = jc(1.0); R[5] = aa[0];
        R[6] = aa[1]; R[7] = jneg(aa[0]); R[8] = jc(1.0);
    }
}

#define RM(R, r, c) (R)[(c) * 3 + (r)]

/* ceres::RotationMatrixToAngleAxis (column-major R): through the quaternion */
static void matrix_to_aa(const jet R[9], jet aa[3])
{
    jet q[4];
    const jet trace = jadd(jadd(RM(R, 0, 0), RM(R, 1, 1)), RM(R, 2, 2));
    if (trace.a >= 0.0) {
        jet t = jsqrt(jadd(trace, jc(1.0)));
        q[0] = jscale(t, 0.5);
        t = jdiv(jc(0.5), t);
        q[1] = jmul(jsub(RM(R, 2, 1), RM(R, 1, 2)), t);
        q[2] = jmul(jsub(RM(R, 0, 2), RM(R, 2, 0)), t);
        q[3] = jmul(jsub(RM(R, 1, 0), RM(R, 0, 1)), t);
    } else {
        int i = 0;
        if (RM(R, 1, 1).a > RM(R, 0, 0).a) i = 1;
        if (RM(R, 2, 2).a > RM(R, i, i).a) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        jet t = jsqrt(jadd(jsub(jsub(RM(R, i, i), RM(R, j, j)), RM(R, k, k)), jc(1.0)));
        q[i + 1] = jscale(t, 0.5);
        t = jdiv(jc(0.5), t);
        q[0] = jmul(jsub(RM(R, k, j), RM(R, j, k)), t);
        q[j + 1] = jmul(jadd(RM(R, j, i), RM(R, i, j)), t);
        q[k + 1] = jmul(jadd(RM(R, k, i), RM(R, i, k)), t);
    }
    /* QuaternionToAngleAxis */
    const jet s2 = jadd(jadd(jmul(q[1], q[1]), jmul(q[2], q[2])), jmul(q[3], q[3]));
    if (s2.a > 0.0) {
        const jet s = jsqrt(s2);
        const jet c = q[0];
        const jet two_theta = jscale((c.a < 0.0) ? jatan2(jneg(s), jneg(c)) : jatan2(s, c), 2.0);
        const jet k = jdiv(two_theta, s);
        for (int a = 0; a < 3; a++) aa[a] = jmul(q[a + 1], k);
    } else {
        for (int a = 0; a < 3; a++) aa[a] = jscale(q[a + 1], 2.0);
    }
}

/* 3x3 products on column-major jets: C = A B, C = A^T B, C = A B^T */
static void mm(const jet* A, const jet* B, jet* C, int ta, int tb)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            jet s = jc(0.0);
            for (int k = 0; k < 3; k++) {
                const jet a = ta ? RM(A, k, r) : RM(A, r, k);
                const jet b = tb ? RM(B, c, k) : RM(B, k, c);
                s = (k == 0) ? jmul(a, b) : jadd(s, jmul(a, b));
            }
            RM(C, r, c) = s;
        }
}
static void mv(const jet* A, const jet v[3], jet out[3])
{
    for (int r = 0; r < 3; r++) out[r] = jadd(jadd(jmul(RM(A, r, 0), v[0]), jmul(RM(A, r, 1), v[1])), jmul(RM(A, r, 2), v[2]));
}

