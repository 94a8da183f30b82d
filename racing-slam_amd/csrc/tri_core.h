// tri_core.h — the two-view DLT + gates of triangulation::triangulate_points for ONE correspondence
// (reference src/Triangulation.cpp:50-100, cv::triangulatePoints at :64), shared by K4 (triangulate.hip)
// and K6 (tracks.hip).  Include only in translation units built with -ffp-contract=off: the f64
// rotations and the f32 gates then execute the oracle's IEEE operations one for one.
#pragma once
#include "common.h"

struct TriParams {
    float fx, fy, cx, cy;
    float min_parallax_cosine, max_reprojection_error;
};

__device__ __forceinline__ double cv_hypot(double a, double b)
{
    a = fabs(a);
    b = fabs(b);
    if (a > b) { b /= a; return a * sqrt(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
    return 0;
}

__device__ __forceinline__ float dot3f(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

__device__ __forceinline__ void normalize3f(float* v)
{
    const float n = dot3f(v, v);
    if (n > 0.0f) {
        const float s = sqrtf(n);
        v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s;
    }
}

__device__ __forceinline__ float det3f(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}

__device__ __forceinline__ void inverse_translation(const float* T, float* c)
{
    const float M0 = det3f(T[1], T[2], T[3], T[5], T[6], T[7], T[9], T[10], T[11]);
    const float M1 = det3f(T[0], T[2], T[3], T[4], T[6], T[7], T[8], T[10], T[11]);
    const float M2 = det3f(T[0], T[1], T[3], T[4], T[5], T[7], T[8], T[9], T[11]);
    const float M3 = det3f(T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]);
    const float det = (-T[12] * M0 + T[13] * M1) + (-T[14] * M2 + T[15] * M3);
    c[0] = -M0 / det;
    c[1] = M1 / det;
    c[2] = -M2 / det;
}

__device__ __forceinline__ void projection_rows(const TriParams& k, const float* T, float* P)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        P[0 * 4 + j] = (k.fx * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + k.cx * T[2 * 4 + j];
        P[1 * 4 + j] = (0.0f * T[0 * 4 + j] + k.fy * T[1 * 4 + j]) + k.cy * T[2 * 4 + j];
        P[2 * 4 + j] = (0.0f * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + 1.0f * T[2 * 4 + j];
    }
}

// One Jacobi rotation of rows i, j of At / Vt (static indices after unrolling).
#define JACOBI_PAIR(i, j)                                                                      \
    {                                                                                          \
        double a = W[i], p = 0, b = W[j];                                                      \
        _Pragma("unroll") for (int k = 0; k < 4; k++) p += At[i][k] * At[j][k];                \
        if (!(fabs(p) <= eps * sqrt(a * b))) {                                                 \
            p *= 2;                                                                            \
            const double beta = a - b, gamma = cv_hypot(p, beta);                              \
            double c, s;                                                                       \
            if (beta < 0) {                                                                    \
                const double delta = (gamma - beta) * 0.5;                                     \
                s = sqrt(delta / gamma);                                                       \
                c = p / (gamma * s * 2);                                                       \
            } else {                                                                           \
                c = sqrt((gamma + beta) / (gamma * 2));                                        \
                s = p / (gamma * c * 2);                                                       \
            }                                                                                  \
            a = 0; b = 0;                                                                      \
            _Pragma("unroll") for (int k = 0; k < 4; k++) {                                    \
                const double t0 = c * At[i][k] + s * At[j][k];                                 \
                const double t1 = -s * At[i][k] + c * At[j][k];                                \
                At[i][k] = t0; At[j][k] = t1;                                                  \
                a += t0 * t0; b += t1 * t1;                                                    \
            }                                                                                  \
            W[i] = a; W[j] = b;                                                                \
            changed = true;                                                                    \
            _Pragma("unroll") for (int k = 0; k < 4; k++) {                                    \
                const double t0 = c * Vt[i][k] + s * Vt[j][k];                                 \
                const double t1 = -s * Vt[i][k] + c * Vt[j][k];                                \
                Vt[i][k] = t0; Vt[j][k] = t1;                                                  \
            }                                                                                  \
        }                                                                                      \
    }

__device__ __forceinline__ void load_pose(const float* __restrict__ poses, int idx, float* T)
{
    const float4* a = (const float4*)(poses + 16 * (size_t)idx);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float4 v = a[r];
        T[4 * r] = v.x; T[4 * r + 1] = v.y; T[4 * r + 2] = v.z; T[4 * r + 3] = v.w;
    }
}

// Camera::project, src/Camera.cpp:25-32: K * pose.block<3,4> first, then * homogeneous
__device__ __forceinline__ float2 project_f32(const TriParams& k, const float* T, const float* X)
{
    float KP[12];
    projection_rows(k, T, KP);
    float uvw[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
        uvw[i] = (KP[4 * i] * X[0] + KP[4 * i + 1] * X[1]) + (KP[4 * i + 2] * X[2] + KP[4 * i + 3] * 1.0f);
    if (uvw[2] < 0.0f) return make_float2(-1.0f, -1.0f);
    return make_float2(uvw[0] / uvw[2], uvw[1] / uvw[2]);
}

// -R^T t, src/MotionModel.cpp:8-11, src/Frame.cpp:39-42
__device__ __forceinline__ void camera_center_f32(const float* T, float* c)
{
    const float t[3] = {T[3], T[7], T[11]};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float a[3] = {-T[0 * 4 + i], -T[1 * 4 + i], -T[2 * 4 + i]};
        c[i] = dot3f(a, t);
    }
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One correspondence: pixels p1 / p2 seen under world->camera poses T1 / T2 (row-major 4x4).
// Returns the keep flag; Xout = the f32 point (written even when the gates reject it).
__device__ __forceinline__ bool dlt_one(const float2 p1, const float2 p2, const float* T1, const float* T2,
                                        const TriParams& prm, float* Xout)
{
    double At[4][4], Vt[4][4], W[4];
    {
        float P1[12], P2[12];
        projection_rows(prm, T1, P1);
        projection_rows(prm, T2, P2);
        // A rows: x*P[2]-P[0], y*P[2]-P[1] per view; At[k][row] = A[row][k]
#pragma unroll
        for (int k = 0; k < 4; k++) {
            At[k][0] = (double)p1.x * (double)P1[8 + k] - (double)P1[k];
            At[k][1] = (double)p1.y * (double)P1[8 + k] - (double)P1[4 + k];
            At[k][2] = (double)p2.x * (double)P2[8 + k] - (double)P2[k];
            At[k][3] = (double)p2.y * (double)P2[8 + k] - (double)P2[4 + k];
        }
    }
    const double eps = 2.220446049250313e-16 * 10;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += At[r][k] * At[r][k];
        W[r] = sd;
#pragma unroll
        for (int k = 0; k < 4; k++) Vt[r][k] = (r == k) ? 1.0 : 0.0;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        JACOBI_PAIR(0, 1) JACOBI_PAIR(0, 2) JACOBI_PAIR(0, 3)
        JACOBI_PAIR(1, 2) JACOBI_PAIR(1, 3) JACOBI_PAIR(2, 3)
        if (!changed) break;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += At[r][k] * At[r][k];
        W[r] = sqrt(sd);
    }
    // descending selection sort of the singular values, V^T rows follow; only
    // the row that ends up last (smallest) is needed.
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int j = a;
#pragma unroll
        for (int k = a + 1; k < 4; k++)
            if (W[j] < W[k]) j = k;
        // static-index swap of rows a and j
#pragma unroll
        for (int k = a + 1; k < 4; k++) {
            if (j == k) {
                const double tw = W[a]; W[a] = W[k]; W[k] = tw;
#pragma unroll
                for (int m = 0; m < 4; m++) { const double tv = Vt[a][m]; Vt[a][m] = Vt[k][m]; Vt[k][m] = tv; }
            }
        }
    }
    const float h0 = (float)Vt[3][0], h1 = (float)Vt[3][1], h2 = (float)Vt[3][2], h3 = (float)Vt[3][3];
    float X[3] = {h0 / h3, h1 / h3, h2 / h3};                         // :69-72
    Xout[0] = X[0]; Xout[1] = X[1]; Xout[2] = X[2];

    bool ok = true;
    float c1[3], c2[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {                                      // :74-75
        c1[r] = (T1[4 * r] * X[0] + T1[4 * r + 1] * X[1]) + (T1[4 * r + 2] * X[2] + T1[4 * r + 3] * 1.0f);
        c2[r] = (T2[4 * r] * X[0] + T2[4 * r + 1] * X[1]) + (T2[4 * r + 2] * X[2] + T2[4 * r + 3] * 1.0f);
    }
    if (c1[2] < 0.0f || c2[2] < 0.0f) ok = false;                      // :78
    float o1[3], o2[3];
    inverse_translation(T1, o1);                                       // :83-84
    inverse_translation(T2, o2);
    float a[3] = {o1[0] - X[0], o1[1] - X[1], o1[2] - X[2]};
    float b[3] = {o2[0] - X[0], o2[1] - X[1], o2[2] - X[2]};
    normalize3f(a);
    normalize3f(b);
    if (dot3f(a, b) > prm.min_parallax_cosine) ok = false;            // :86-88
    const float w1 = (0.0f * c1[0] + 0.0f * c1[1]) + 1.0f * c1[2];
    const float w2 = (0.0f * c2[0] + 0.0f * c2[1]) + 1.0f * c2[2];
    const float i1x = ((prm.fx * c1[0] + 0.0f * c1[1]) + prm.cx * c1[2]) / w1;
    const float i1y = ((0.0f * c1[0] + prm.fy * c1[1]) + prm.cy * c1[2]) / w1;
    const float i2x = ((prm.fx * c2[0] + 0.0f * c2[1]) + prm.cx * c2[2]) / w2;
    const float i2y = ((0.0f * c2[0] + prm.fy * c2[1]) + prm.cy * c2[2]) / w2;
    const float e1x = i1x - p1.x, e1y = i1y - p1.y, e2x = i2x - p2.x, e2y = i2y - p2.y;
    const float err1 = sqrtf(e1x * e1x + e1y * e1y);                   // :95-96
    const float err2 = sqrtf(e2x * e2x + e2y * e2y);
    if (err1 > prm.max_reprojection_error || err2 > prm.max_reprojection_error) ok = false;   // :97-100
    return ok;
}
