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

// ---------------------------------------------------------------------------------------- host in, host out
// triangulate_points as the reference's callers see it: std::vectors in, std::vector out, and — in Mapper::triangulate_tracks
// (src/Mapper.cpp:253) and pose::recover_pose (src/PoseEstimation.cpp:48) — a handful of correspondences per call.  Through
// the staging pool such a call is 3 uploads + 2 launches + 3 downloads + a stream synchronisation (50 us for n = 1).  Up to
// RS_TRI_SMALL correspondences take ONE launch of ONE workgroup instead: the two poses travel as kernel arguments, the
// correspondences are read straight from a pinned host block, the kept points are written — compacted, in input order — into
// that block, and the last instruction raises a completion flag the host spins on.  No copy launch, no hipStreamSynchronize.
#define RS_TRI_SMALL 256
struct TriPoses { float T1[16], T2[16]; };
struct TriSmallOut { volatile int flag; int count; int pad[2]; int32_t index[RS_TRI_SMALL]; float xyz[RS_TRI_SMALL * 3]; };

__global__ __launch_bounds__(RS_TRI_SMALL) void k4_small(const float2* __restrict__ uv /*pinned: [n] view 1, then [n] view 2*/, int n,
                                                         float2 a0, float2 b0, TriPoses ps, TriParams prm, TriSmallOut* __restrict__ out, int ticket)
{
    const int i = threadIdx.x;
    bool ok = false;
    float X[3] = {0.f, 0.f, 0.f};
    if (i < n) {
        const float2 p1 = n == 1 ? a0 : uv[i], p2 = n == 1 ? b0 : uv[n + i];      // (a single correspondence is a kernel argument)
        ok = dlt_one(p1, p2, ps.T1, ps.T2, prm, X);
    }
    int total;
    const int off = rs_block_exclusive_scan(ok ? 1 : 0, &total);
    if (ok) {
        out->index[off] = i;
        out->xyz[3 * off] = X[0]; out->xyz[3 * off + 1] = X[1]; out->xyz[3 * off + 2] = X[2];
    }
    if (i == 0) out->count = total;
    __threadfence_system();
    __syncthreads();
    if (i == 0) out->flag = ticket;
}

extern "C" int rs_triangulate_host(rs_context* ctx, const float* h_uv1, const float* h_uv2, int n, const float h_pose1[16],
                                   const float h_pose2[16], const float h_intrinsics[4], float min_parallax_cosine,
                                   float max_reprojection_error, int32_t* h_out_index, float* h_out_xyz, int* h_count)
{
    if (!ctx || !h_count) return RS_ERR_INVALID;
    *h_count = 0;
    if (n < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative n");
    if (n == 0) return RS_OK;                                       // empty guard, src/Triangulation.cpp:46-48
    if (!h_uv1 || !h_uv2 || !h_pose1 || !h_pose2 || !h_intrinsics || !h_out_index || !h_out_xyz) return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    if (n > RS_TRI_SMALL) {
        // a whole frame pair: staging pool + the grid kernels
        int rc = rs_stage_begin(ctx);
        if (rc) return rc;
        float *d1 = nullptr, *d2 = nullptr, *dp = nullptr, *xyz = nullptr, *oxyz = nullptr;
        uint8_t* keep = nullptr;
        int32_t *oidx = nullptr, *cnt = nullptr;
        float poses[32];
        memcpy(poses, h_pose1, sizeof(float) * 16);
        memcpy(poses + 16, h_pose2, sizeof(float) * 16);
        const size_t N = (size_t)n;
        if ((rc = rs_stage_upload(ctx, h_uv1, sizeof(float) * 2 * N, (void**)&d1))) return rc;
        if ((rc = rs_stage_upload(ctx, h_uv2, sizeof(float) * 2 * N, (void**)&d2))) return rc;
        if ((rc = rs_stage_upload(ctx, poses, sizeof poses, (void**)&dp))) return rc;
        if ((rc = rs_stage_alloc(ctx, sizeof(float) * 3 * N, (void**)&xyz))) return rc;
        if ((rc = rs_stage_alloc(ctx, sizeof(float) * 3 * N, (void**)&oxyz))) return rc;
        if ((rc = rs_stage_alloc(ctx, N, (void**)&keep))) return rc;
        if ((rc = rs_stage_alloc(ctx, sizeof(int32_t) * N, (void**)&oidx))) return rc;
        if ((rc = rs_stage_alloc(ctx, sizeof(int32_t), (void**)&cnt))) return rc;
        rc = rs_triangulate(ctx, d1, d2, n, dp, 2, nullptr, nullptr, h_intrinsics, min_parallax_cosine, max_reprojection_error, xyz, keep, oidx, oxyz, cnt);
        if (rc) return rc;
        int32_t m = 0;
        if ((rc = rs_stage_download(ctx, cnt, sizeof(int32_t), &m))) return rc;
        if ((rc = rs_stage_download(ctx, oidx, sizeof(int32_t) * N, h_out_index))) return rc;
        if ((rc = rs_stage_download(ctx, oxyz, sizeof(float) * 3 * N, h_out_xyz))) return rc;
        if ((rc = rs_stage_sync(ctx))) return rc;
        *h_count = m;
        return RS_OK;
    }
    if (!ctx->tri_pin) {
        if (hipHostMalloc(&ctx->tri_pin, sizeof(TriSmallOut) + sizeof(float2) * 2 * RS_TRI_SMALL, hipHostMallocDefault) != hipSuccess)
            return rs_fail(ctx, RS_ERR_NOMEM, "pinned block of the small triangulation");
        ((TriSmallOut*)ctx->tri_pin)->flag = 0;
    }
    TriSmallOut* out = (TriSmallOut*)ctx->tri_pin;
    float2* uv = (float2*)(out + 1);
    TriPoses ps;
    memcpy(ps.T1, h_pose1, sizeof ps.T1);
    memcpy(ps.T2, h_pose2, sizeof ps.T2);
    float2 a0 = make_float2(h_uv1[0], h_uv1[1]), b0 = make_float2(h_uv2[0], h_uv2[1]);
    if (n > 1) {
        memcpy(uv, h_uv1, sizeof(float2) * (size_t)n);
        memcpy(uv + n, h_uv2, sizeof(float2) * (size_t)n);
    }
    const TriParams prm = {h_intrinsics[0], h_intrinsics[1], h_intrinsics[2], h_intrinsics[3], min_parallax_cosine, max_reprojection_error};
    const int ticket = ++ctx->tri_ticket;
    {
        rs_prof_scope psx(ctx, "K4s_triangulate_small");
        hipLaunchKernelGGL(k4_small, dim3(1), dim3(RS_TRI_SMALL), 0, ctx->stream, (const float2*)uv, n, a0, b0, ps, prm, out, ticket);
    }
    RS_HIP(ctx, hipGetLastError());
    long spins = 0;
    while (out->flag != ticket) {
        if ((++spins & 0x3FFFF) == 0 && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;      // the stream drained without the flag: a failed launch
    }
    if (out->flag != ticket) RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (out->flag != ticket) return rs_fail(ctx, RS_ERR_HIP, "small triangulation did not complete");
    const int m = out->count;
    memcpy(h_out_index, out->index, sizeof(int32_t) * (size_t)m);
    memcpy(h_out_xyz, out->xyz, sizeof(float) * 3 * (size_t)m);
    *h_count = m;
    return RS_OK;
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
