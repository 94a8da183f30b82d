// triangulate.hip — K4: two-view DLT triangulation + gates, one lane per
// correspondence, then an ordered compaction.
//
// Replaces triangulation::triangulate_points (reference
// src/Triangulation.cpp:37-106) including the cv::triangulatePoints call at :64
// and the per-track call pattern of Mapper::triangulate_tracks
// (src/Mapper.cpp:246-259) through per-item pose indices.
//
// The 4x4 one-sided Jacobi SVD runs entirely in registers in f64 (the matrix
// and V^T are 32 doubles; the pair loops are fully unrolled so every index is
// static).  Built with -ffp-contract=off: the f64 rotations and the f32 gates
// then execute the same IEEE operations, in the same order, as the oracle
// (sqrt and division are correctly rounded on gfx950), so positions and keep
// flags can be compared bit for bit.
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

__device__ __forceinline__ void k4_triangulate_body(
    const float2* __restrict__ uv1, const float2* __restrict__ uv2, int n,
    const float* __restrict__ poses, const int32_t* __restrict__ idx1, const int32_t* __restrict__ idx2,
    const int32_t* __restrict__ gather1, const int32_t* __restrict__ gather2, const int32_t* __restrict__ d_n,
    TriParams prm, float* __restrict__ xyz, uint8_t* __restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (d_n) n = min(n, *d_n);          // match count produced on the device (K1b)
    if (i >= n) return;
    float T1[16], T2[16];
    {
        const float4* a = (const float4*)(poses + 16 * (size_t)(idx1 ? idx1[i] : 0));
        const float4* b = (const float4*)(poses + 16 * (size_t)(idx2 ? idx2[i] : 1));
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float4 va = a[r], vb = b[r];
            T1[4 * r] = va.x; T1[4 * r + 1] = va.y; T1[4 * r + 2] = va.z; T1[4 * r + 3] = va.w;
            T2[4 * r] = vb.x; T2[4 * r + 1] = vb.y; T2[4 * r + 2] = vb.z; T2[4 * r + 3] = vb.w;
        }
    }
    // get_matching_points (src/Triangulation.cpp:11-26) fused: optional gather through the match list
    const float2 p1 = uv1[gather1 ? gather1[i] : i], p2 = uv2[gather2 ? gather2[i] : i];
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
    xyz[3 * (size_t)i + 0] = X[0];
    xyz[3 * (size_t)i + 1] = X[1];
    xyz[3 * (size_t)i + 2] = X[2];

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
    keep[i] = ok ? 1 : 0;
}

// Ordered compaction of the kept correspondences (:102) by ONE workgroup of any size.
__device__ __forceinline__ void k4_compact_body(const uint8_t* __restrict__ keep, const float* __restrict__ xyz,
                                                int n, const int32_t* __restrict__ d_n,
                                                int32_t* __restrict__ out_index, float* __restrict__ out_xyz,
                                                int32_t* __restrict__ out_count)
{
    if (d_n) n = min(n, *d_n);
    // every thread owns a contiguous chunk: count, one workgroup scan, ordered write.  Loads go out in
    // batches (clamped addresses, no branches) so that their latencies overlap.
    const int T = blockDim.x, chunk = (n + T - 1) / T;
    const int lo = min((int)threadIdx.x * chunk, n), hi = min(lo + chunk, n);
    int cnt = 0;
    for (int i0 = lo; i0 < hi; i0 += 16) {
        uint8_t k[16];
#pragma unroll
        for (int u = 0; u < 16; u++) k[u] = keep[min(i0 + u, hi - 1)];
#pragma unroll
        for (int u = 0; u < 16; u++) cnt += (i0 + u < hi && k[u] != 0) ? 1 : 0;
    }
    int total;
    int off = rs_block_exclusive_scan(cnt, &total);
    for (int i0 = lo; i0 < hi; i0 += 8) {
        uint8_t k[8];
        float x[8][3];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = min(i0 + u, hi - 1);
            k[u] = keep[i];
            x[u][0] = xyz[3 * (size_t)i + 0]; x[u][1] = xyz[3 * (size_t)i + 1]; x[u][2] = xyz[3 * (size_t)i + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (i0 + u >= hi || k[u] == 0) continue;
            out_index[off] = i0 + u;
            out_xyz[3 * (size_t)off + 0] = x[u][0];
            out_xyz[3 * (size_t)off + 1] = x[u][1];
            out_xyz[3 * (size_t)off + 2] = x[u][2];
            off++;
        }
    }
    if (threadIdx.x == 0) *out_count = total;
}

// K4 + K4b: every workgroup triangulates 64 correspondences; the last one to finish compacts
// (a separate single-workgroup launch costs ~4 us).
__global__ __launch_bounds__(64) void k4_triangulate(
    const float2* __restrict__ uv1, const float2* __restrict__ uv2, int n,
    const float* __restrict__ poses, const int32_t* __restrict__ idx1, const int32_t* __restrict__ idx2,
    const int32_t* __restrict__ gather1, const int32_t* __restrict__ gather2, const int32_t* __restrict__ d_n,
    TriParams prm, float* __restrict__ xyz, uint8_t* __restrict__ keep, int* ticket,
    int32_t* __restrict__ out_index, float* __restrict__ out_xyz, int32_t* __restrict__ out_count)
{
    k4_triangulate_body(uv1, uv2, n, poses, idx1, idx2, gather1, gather2, d_n, prm, xyz, keep);
    if (rs_last_workgroup(ticket)) k4_compact_body(keep, xyz, n, d_n, out_index, out_xyz, out_count);
}

static int tri_launch(rs_context* ctx, const float* d_uv1, const float* d_uv2, int n, const float* d_poses,
                      int n_poses, const int32_t* d_pose_idx1, const int32_t* d_pose_idx2,
                      const int32_t* g1, const int32_t* g2, const int32_t* d_n, const float h_intrinsics[4],
                      float min_parallax_cosine, float max_reprojection_error, float* d_xyz, uint8_t* d_keep,
                      int32_t* d_out_index, float* d_out_xyz, int32_t* d_out_count)
{
    if (!ctx) return RS_ERR_INVALID;
    if (n < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative n");
    if (!d_out_count) return rs_fail(ctx, RS_ERR_INVALID, "null out_count");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) {   // empty guard, src/Triangulation.cpp:46-48
        RS_HIP(ctx, hipMemsetAsync(d_out_count, 0, sizeof(int32_t), ctx->stream));
        return RS_OK;
    }
    if (!d_uv1 || !d_uv2 || !d_poses || !h_intrinsics || !d_xyz || !d_keep || !d_out_index || !d_out_xyz)
        return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    if ((!d_pose_idx1 || !d_pose_idx2) && n_poses < 2) return rs_fail(ctx, RS_ERR_INVALID, "need two poses");
    if ((uintptr_t)d_poses & 15) return rs_fail(ctx, RS_ERR_INVALID, "poses must be 16-byte aligned");
    TriParams prm = {h_intrinsics[0], h_intrinsics[1], h_intrinsics[2], h_intrinsics[3], min_parallax_cosine,
                     max_reprojection_error};
    {
        rs_prof_scope ps(ctx, "K4_triangulate_dlt");
        hipLaunchKernelGGL(k4_triangulate, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, (const float2*)d_uv1,
                           (const float2*)d_uv2, n, d_poses, d_pose_idx1, d_pose_idx2, g1, g2, d_n, prm, d_xyz, d_keep,
                           ctx->tickets + RS_TICKET_K4, d_out_index, d_out_xyz, d_out_count);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}

extern "C" int rs_triangulate(rs_context* ctx, const float* d_uv1, const float* d_uv2, int n,
                              const float* d_poses, int n_poses, const int32_t* d_pose_idx1,
                              const int32_t* d_pose_idx2, const float h_intrinsics[4],
                              float min_parallax_cosine, float max_reprojection_error, float* d_xyz,
                              uint8_t* d_keep, int32_t* d_out_index, float* d_out_xyz, int32_t* d_out_count)
{
    return tri_launch(ctx, d_uv1, d_uv2, n, d_poses, n_poses, d_pose_idx1, d_pose_idx2, nullptr, nullptr, nullptr,
                      h_intrinsics, min_parallax_cosine, max_reprojection_error, d_xyz, d_keep, d_out_index,
                      d_out_xyz, d_out_count);
}

extern "C" int rs_triangulate_matches(rs_context* ctx, const float* d_kp1, const float* d_kp2,
                                      const int32_t* d_match_train, const int32_t* d_match_query,
                                      const int32_t* d_n_matches, int max_matches, const float* d_poses,
                                      const float h_intrinsics[4], float min_parallax_cosine,
                                      float max_reprojection_error, float* d_xyz, uint8_t* d_keep,
                                      int32_t* d_out_index, float* d_out_xyz, int32_t* d_out_count)
{
    if (ctx && max_matches > 0 && (!d_match_train || !d_match_query || !d_n_matches))
        return rs_fail(ctx, RS_ERR_INVALID, "null match list");
    return tri_launch(ctx, d_kp1, d_kp2, max_matches, d_poses, 2, nullptr, nullptr, d_match_train, d_match_query,
                      d_n_matches, h_intrinsics, min_parallax_cosine, max_reprojection_error, d_xyz, d_keep,
                      d_out_index, d_out_xyz, d_out_count);
}
