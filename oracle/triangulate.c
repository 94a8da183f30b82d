/*
 * oracle/triangulate.c — CPU restatement of triangulation::triangulate_points
 * (reference src/Triangulation.cpp:37-106; projection matrix src/Camera.cpp:47-57).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see rs_oracle.h).
 *
 * Third-party semantics restated (OpenCV 4.x calib3d + core, not in
 * /root/reference): cv::triangulatePoints builds, per correspondence, the
 * 4x4 f64 matrix with rows  x*P[2]-P[0], y*P[2]-P[1]  (view 1 then view 2),
 * runs cv::SVD (one-sided Jacobi on the columns, eps = 10*DBL_EPSILON,
 * <= 30 sweeps, OpenCV's own scaled hypot), sorts the singular values in
 * descending order and returns the last row of V^T, stored as f32.
 *
 * f32 gate arithmetic follows the operation order of reproj_match.c
 * (dot3/dot4 trees, no FMA contraction); pose.inverse() (:83-84) is restated
 * as the cofactor inverse restricted to the translation column.
 */
#include <math.h>

#include "rs_oracle.h"

static double cv_hypot(double a, double b)
{
    a = fabs(a);
    b = fabs(b);
    if (a > b) { b /= a; return a * sqrt(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
    return 0;
}

/* At holds A transposed (row i of At = column i of A).  Vt rows accumulate the
 * right singular vectors. */
void orc_null_vector4(const double A[16], double v[4], double sigma[4])
{
    double At[4][4], Vt[4][4], W[4];
    const double eps = 2.220446049250313e-16 * 10;
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++) At[i][k] = A[k * 4 + i];
    for (int i = 0; i < 4; i++) {
        double sd = 0;
        for (int k = 0; k < 4; k++) sd += At[i][k] * At[i][k];
        W[i] = sd;
        for (int k = 0; k < 4; k++) Vt[i][k] = (i == k) ? 1.0 : 0.0;
    }
    for (int iter = 0; iter < 30; iter++) {
        int changed = 0;
        for (int i = 0; i < 3; i++) {
            for (int j = i + 1; j < 4; j++) {
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 4; k++) p += At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = cv_hypot(p, beta), c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = 0; b = 0;
                for (int k = 0; k < 4; k++) {
                    double t0 = c * At[i][k] + s * At[j][k];
                    double t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                for (int k = 0; k < 4; k++) {
                    double t0 = c * Vt[i][k] + s * Vt[j][k];
                    double t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0; Vt[j][k] = t1;
                }
            }
        }
        if (!changed) break;
    }
    for (int i = 0; i < 4; i++) {
        double sd = 0;
        for (int k = 0; k < 4; k++) sd += At[i][k] * At[i][k];
        W[i] = sqrt(sd);
    }
    /* descending selection sort, swapping the V^T rows along */
    for (int i = 0; i < 3; i++) {
        int j = i;
        for (int k = i + 1; k < 4; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (int k = 0; k < 4; k++) { t = Vt[i][k]; Vt[i][k] = Vt[j][k]; Vt[j][k] = t; }
        }
    }
    for (int k = 0; k < 4; k++) { v[k] = Vt[3][k]; sigma[k] = W[k]; }
}

static float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

static void normalize3(float* v)
{
    float n = dot3(v, v);
    if (n > 0.0f) {
        float s = sqrtf(n);
        v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s;
    }
}

static float det3(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}

/* translation column of the general 4x4 inverse: inv(i,3) = (-1)^(3+i) M(3,i) / det */
static void inverse_translation(const float* T, float c[3])
{
    /* minors deleting row 3 and column j */
    float M0 = det3(T[1], T[2], T[3], T[5], T[6], T[7], T[9], T[10], T[11]);
    float M1 = det3(T[0], T[2], T[3], T[4], T[6], T[7], T[8], T[10], T[11]);
    float M2 = det3(T[0], T[1], T[3], T[4], T[5], T[7], T[8], T[9], T[11]);
    float M3 = det3(T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]);
    float det = (-T[12] * M0 + T[13] * M1) + (-T[14] * M2 + T[15] * M3);
    c[0] = -M0 / det;
    c[1] = M1 / det;
    c[2] = -M2 / det;
}

static void projection(const float* K, const float* T, float* P)
{
    /* K * pose.block<3,4>(0,0), src/Camera.cpp:49 */
    for (int j = 0; j < 4; j++) {
        P[0 * 4 + j] = (K[0] * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + K[2] * T[2 * 4 + j];
        P[1 * 4 + j] = (0.0f * T[0 * 4 + j] + K[1] * T[1 * 4 + j]) + K[3] * T[2 * 4 + j];
        P[2 * 4 + j] = (0.0f * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + 1.0f * T[2 * 4 + j];
    }
}

int orc_triangulate(const float* uv1, const float* uv2, int n, const float* poses, int n_poses,
                    const int32_t* pose_idx1, const int32_t* pose_idx2,
                    const float K[4], float min_parallax_cosine,
                    float max_reprojection_error, float* xyz, uint8_t* keep,
                    int32_t* out_index, float* out_xyz, int32_t* out_count)
{
    *out_count = 0;
    if (n <= 0) return 0;                                   /* src/Triangulation.cpp:46-48 */
    if (n_poses < 2 && !(pose_idx1 && pose_idx2)) return 1;
    /* per-correspondence work (independent; parallel in the all-cores baseline build), ordered compaction after */
#ifdef ORC_OMP
#pragma omp parallel for schedule(static, 64)
#endif
    for (int i = 0; i < n; i++) {
        const float* T1 = poses + 16 * (size_t)(pose_idx1 ? pose_idx1[i] : 0);
        const float* T2 = poses + 16 * (size_t)(pose_idx2 ? pose_idx2[i] : 1);
        float P1[12], P2[12];
        projection(K, T1, P1);
        projection(K, T2, P2);
        double A[16];
        const float* Ps[2] = {P1, P2};
        const float* uvs[2] = {uv1 + 2 * (size_t)i, uv2 + 2 * (size_t)i};
        for (int j = 0; j < 2; j++) {
            double x = uvs[j][0], y = uvs[j][1];
            for (int k = 0; k < 4; k++) {
                A[(2 * j + 0) * 4 + k] = x * (double)Ps[j][8 + k] - (double)Ps[j][k];
                A[(2 * j + 1) * 4 + k] = y * (double)Ps[j][8 + k] - (double)Ps[j][4 + k];
            }
        }
        double v[4], sg[4];
        orc_null_vector4(A, v, sg);
        float h[4] = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};  /* CV_32F 4xN */
        float X[3] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};               /* :69-72 */
        xyz[3 * i + 0] = X[0]; xyz[3 * i + 1] = X[1]; xyz[3 * i + 2] = X[2];
        keep[i] = 0;

        float c1[3], c2[3];
        for (int r = 0; r < 3; r++) {                                        /* :74-75 */
            c1[r] = (T1[4 * r] * X[0] + T1[4 * r + 1] * X[1]) + (T1[4 * r + 2] * X[2] + T1[4 * r + 3] * 1.0f);
            c2[r] = (T2[4 * r] * X[0] + T2[4 * r + 1] * X[1]) + (T2[4 * r + 2] * X[2] + T2[4 * r + 3] * 1.0f);
        }
        if (c1[2] < 0.0f || c2[2] < 0.0f) continue;                          /* :78 */

        float o1[3], o2[3];
        inverse_translation(T1, o1);                                         /* :83-84 */
        inverse_translation(T2, o2);
        float a[3] = {o1[0] - X[0], o1[1] - X[1], o1[2] - X[2]};
        float b[3] = {o2[0] - X[0], o2[1] - X[1], o2[2] - X[2]};
        normalize3(a);
        normalize3(b);
        float similarity = dot3(a, b);
        if (similarity > min_parallax_cosine) continue;                      /* :86-88 */

        /* (K * cam_point).hnormalized(), :91-92 */
        float w1 = (0.0f * c1[0] + 0.0f * c1[1]) + 1.0f * c1[2];
        float w2 = (0.0f * c2[0] + 0.0f * c2[1]) + 1.0f * c2[2];
        float i1x = ((K[0] * c1[0] + 0.0f * c1[1]) + K[2] * c1[2]) / w1;
        float i1y = ((0.0f * c1[0] + K[1] * c1[1]) + K[3] * c1[2]) / w1;
        float i2x = ((K[0] * c2[0] + 0.0f * c2[1]) + K[2] * c2[2]) / w2;
        float i2y = ((0.0f * c2[0] + K[1] * c2[1]) + K[3] * c2[2]) / w2;
        float e1x = i1x - uvs[0][0], e1y = i1y - uvs[0][1];
        float e2x = i2x - uvs[1][0], e2y = i2y - uvs[1][1];
        float err1 = sqrtf(e1x * e1x + e1y * e1y);                           /* :95-96 */
        float err2 = sqrtf(e2x * e2x + e2y * e2y);
        if (err1 > max_reprojection_error || err2 > max_reprojection_error) continue;  /* :97-100 */

        keep[i] = 1;
    }
    int count = 0;
    for (int i = 0; i < n; i++) {
        if (!keep[i]) continue;
        out_index[count] = i;                                                /* :102 */
        out_xyz[3 * count + 0] = xyz[3 * i + 0]; out_xyz[3 * count + 1] = xyz[3 * i + 1]; out_xyz[3 * count + 2] = xyz[3 * i + 2];
        count++;
    }
    *out_count = count;
    return 0;
}
