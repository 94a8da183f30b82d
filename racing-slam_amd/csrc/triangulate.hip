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
#include "tri_core.h"

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
    float X[3];
    const bool ok = dlt_one(p1, p2, T1, T2, prm, X);
    xyz[3 * (size_t)i + 0] = X[0];
    xyz[3 * (size_t)i + 1] = X[1];
    xyz[3 * (size_t)i + 2] = X[2];
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

// K4: every workgroup triangulates 64 correspondences.  (Running the compaction in the last workgroup to finish
// instead of a second launch was measured: release fence + ticket + acquire + a 64-thread tail cost more GPU time
// than the ~4 us launch they saved.)
__global__ __launch_bounds__(64) void k4_triangulate(
    const float2* __restrict__ uv1, const float2* __restrict__ uv2, int n,
    const float* __restrict__ poses, const int32_t* __restrict__ idx1, const int32_t* __restrict__ idx2,
    const int32_t* __restrict__ gather1, const int32_t* __restrict__ gather2, const int32_t* __restrict__ d_n,
    TriParams prm, float* __restrict__ xyz, uint8_t* __restrict__ keep)
{
    k4_triangulate_body(uv1, uv2, n, poses, idx1, idx2, gather1, gather2, d_n, prm, xyz, keep);
}

// K4b: one workgroup, ordered compaction of the kept correspondences (:102)
__global__ __launch_bounds__(1024) void k4_compact(const uint8_t* __restrict__ keep, const float* __restrict__ xyz, int n,
                                                   const int32_t* __restrict__ d_n, int32_t* __restrict__ out_index,
                                                   float* __restrict__ out_xyz, int32_t* __restrict__ out_count)
{
    k4_compact_body(keep, xyz, n, d_n, out_index, out_xyz, out_count);
}

// cfg 4 (BASELINE.json configs[3]): a batch of independent frame pairs in one launch pair.  blockIdx.y = pair; every
// array is the single-pair array with a leading batch dimension.
__global__ __launch_bounds__(64) void k4_triangulate_pairs(
    const float2* __restrict__ kp1, int n1, const float2* __restrict__ kp2, int n2, int stride,
    const float* __restrict__ poses, const int32_t* __restrict__ match_train, const int32_t* __restrict__ match_query,
    const int32_t* __restrict__ d_n, TriParams prm, float* __restrict__ xyz, uint8_t* __restrict__ keep)
{
    const size_t b = blockIdx.y;
    k4_triangulate_body(kp1 + b * n1, kp2 + b * n2, stride, poses + b * 32, nullptr, nullptr, match_train + b * stride,
                        match_query + b * stride, d_n + b, prm, xyz + b * stride * 3, keep + b * stride);
}

__global__ __launch_bounds__(1024) void k4_compact_pairs(const uint8_t* __restrict__ keep, const float* __restrict__ xyz,
                                                         int stride, const int32_t* __restrict__ d_n,
                                                         int32_t* __restrict__ out_index, float* __restrict__ out_xyz,
                                                         int32_t* __restrict__ out_count)
{
    const size_t b = blockIdx.x;
    k4_compact_body(keep + b * stride, xyz + b * stride * 3, stride, d_n + b, out_index + b * stride,
                    out_xyz + b * stride * 3, out_count + b);
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
                           (const float2*)d_uv2, n, d_poses, d_pose_idx1, d_pose_idx2, g1, g2, d_n, prm, d_xyz, d_keep);
    }
    {
        rs_prof_scope ps(ctx, "K4b_compact");
        hipLaunchKernelGGL(k4_compact, dim3(1), dim3(1024), 0, ctx->stream, d_keep, d_xyz, n, d_n, d_out_index,
                           d_out_xyz, d_out_count);
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

extern "C" int rs_triangulate_matches_batch(rs_context* ctx, int batch, const float* d_kp1, int n1, const float* d_kp2,
                                            int n2, const int32_t* d_match_train, const int32_t* d_match_query,
                                            const int32_t* d_n_matches, int max_matches, const float* d_poses,
                                            const float h_intrinsics[4], float min_parallax_cosine,
                                            float max_reprojection_error, float* d_xyz, uint8_t* d_keep,
                                            int32_t* d_out_index, float* d_out_xyz, int32_t* d_out_count)
{
    if (!ctx) return RS_ERR_INVALID;
    if (batch < 0 || n1 < 0 || n2 < 0 || max_matches < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (batch > 65535) return rs_fail(ctx, RS_ERR_UNSUPPORTED, "more than 65535 pairs per call");
    if (batch == 0) return RS_OK;
    if (!d_out_count) return rs_fail(ctx, RS_ERR_INVALID, "null out_count");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    if (max_matches == 0) {   // empty guard, src/Triangulation.cpp:46-48
        RS_HIP(ctx, hipMemsetAsync(d_out_count, 0, sizeof(int32_t) * (size_t)batch, ctx->stream));
        return RS_OK;
    }
    if (!d_kp1 || !d_kp2 || !d_match_train || !d_match_query || !d_n_matches || !d_poses || !h_intrinsics || !d_xyz ||
        !d_keep || !d_out_index || !d_out_xyz)
        return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    if ((uintptr_t)d_poses & 15) return rs_fail(ctx, RS_ERR_INVALID, "poses must be 16-byte aligned");
    TriParams prm = {h_intrinsics[0], h_intrinsics[1], h_intrinsics[2], h_intrinsics[3], min_parallax_cosine,
                     max_reprojection_error};
    {
        rs_prof_scope ps(ctx, "K4_triangulate_dlt");
        hipLaunchKernelGGL(k4_triangulate_pairs, dim3((max_matches + 63) / 64, batch), dim3(64), 0, ctx->stream,
                           (const float2*)d_kp1, n1, (const float2*)d_kp2, n2, max_matches, d_poses, d_match_train,
                           d_match_query, d_n_matches, prm, d_xyz, d_keep);
    }
    {
        rs_prof_scope ps(ctx, "K4b_compact");
        hipLaunchKernelGGL(k4_compact_pairs, dim3(batch), dim3(1024), 0, ctx->stream, d_keep, d_xyz, max_matches,
                           d_n_matches, d_out_index, d_out_xyz, d_out_count);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}
