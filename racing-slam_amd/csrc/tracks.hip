// tracks.hip — K6: the body of Mapper::triangulate_tracks (reference src/Mapper.cpp:246-305) as two
// launches: per-track triangulation + sighting consistency + parallax terms (one lane per track), then the
// selection (threshold, quota top-up) by one workgroup.  SURVEY.md §8(f) rank 1: it replaces ~2000
// one-correspondence cv::triangulatePoints calls per key frame.
//
// Per track (track-id order = input order): triangulate (first sighting, key-frame pixel) with the loose
// gates (:252-262, through dlt_one of tri_core.h), reproject into every sighting's pose, first error > 4 px
// makes the track inconsistent (:264-275), parallax cosine and the rotation-dependent requirement
// (:277-288).  Selection (:291-304): candidates at or below their requirement in order, then — below the
// quota — the best of the rest by parallax cosine ascending (ties: candidate order; the reference's
// std::sort leaves them unspecified).  Built with -ffp-contract=off; acosf / cosf are the device libm's
// (last-ulp differences to glibc only move `required`).
#include "tri_core.h"

struct TrackParams {
    TriParams tri;               // loose gates of the per-track triangulation
    float min_parallax_cosine;   // TRACK_MIN_PARALLAX_COSINE
    float rotation_factor;       // ROTATION_PARALLAX_FACTOR
    int kf_pose;
};

__global__ __launch_bounds__(64) void k6_tracks(const float2* __restrict__ track_uv, const uint8_t* __restrict__ skip,
                                               const int32_t* __restrict__ sight_ptr,
                                               const int32_t* __restrict__ sight_pose,
                                               const float2* __restrict__ sight_uv, const float* __restrict__ poses,
                                               int n_tracks, TrackParams prm, uint8_t* __restrict__ status,
                                               float* __restrict__ xyz, float* __restrict__ parallax_cos,
                                               float* __restrict__ required_cos, const float* __restrict__ required_by_pose)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tracks) return;
    const int s0 = sight_ptr[t], s1 = sight_ptr[t + 1];
    uint8_t st = 0;
    float X[3] = {0.f, 0.f, 0.f}, pc = 0.f, rc = 0.f;
    if (s1 > s0 && !(skip && skip[t])) {                                   // :248-250
        float Tf[16], Tk[16];
        load_pose(poses, sight_pose[s0], Tf);
        load_pose(poses, prm.kf_pose, Tk);
        const bool ok = dlt_one(sight_uv[s0], track_uv[t], Tf, Tk, prm.tri, X);   // :254-263
        if (ok) {
            bool consistent = true;
            for (int s = s0; s < s1; s++) {                                // :265-271
                float Ts[16];
                load_pose(poses, sight_pose[s], Ts);
                const float2 pr = project_f32(prm.tri, Ts, X);
                const float2 px = sight_uv[s];
                const float dx = pr.x - px.x, dy = pr.y - px.y;
                if (sqrtf(dx * dx + dy * dy) > prm.tri.max_reprojection_error) { consistent = false; break; }
            }
            if (!consistent) {
                st = 2;                                                    // :272-275
            } else {
                float cf[3], ck[3];
                camera_center_f32(Tf, cf);
                camera_center_f32(Tk, ck);
                float a[3] = {cf[0] - X[0], cf[1] - X[1], cf[2] - X[2]};
                float b[3] = {ck[0] - X[0], ck[1] - X[1], ck[2] - X[2]};
                normalize3f(a);                                            // :278-279
                normalize3f(b);
                float tr[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {                              // trace(R_kf R_first^T), :281-282
                    const float rk[3] = {Tk[4 * i], Tk[4 * i + 1], Tk[4 * i + 2]};
                    const float rf[3] = {Tf[4 * i], Tf[4 * i + 1], Tf[4 * i + 2]};
                    tr[i] = dot3f(rk, rf);
                }
                const float trace = (tr[0] + tr[1]) + tr[2];
                float cosine = (trace - 1.0f) / 2.0f;
                cosine = cosine < -1.0f ? -1.0f : cosine;
                cosine = cosine > 1.0f ? 1.0f : cosine;
                st = 1;
                pc = dot3f(a, b);                                          // :287
                if (required_by_pose) {
                    rc = required_by_pose[sight_pose[s0]];                 // the host's libm (rs_parallax_requirements): bit-exact
                } else {
                    const float turned = acosf(cosine);
                    const float need = cosf(prm.rotation_factor * turned);
                    rc = prm.min_parallax_cosine < need ? prm.min_parallax_cosine : need;   // :288
                }
            }
        }
    }
    status[t] = st;
    xyz[3 * (size_t)t + 0] = X[0]; xyz[3 * (size_t)t + 1] = X[1]; xyz[3 * (size_t)t + 2] = X[2];
    parallax_cos[t] = pc;
    required_cos[t] = rc;
}

// Selection by ONE workgroup (:291-304): ordered compaction of the accepted and of the inconsistent tracks,
// then the quota top-up by rank counting over the rejected candidates (n_rej^2 comparisons; key = (cosine,
// candidate order), NaN cosines sort last).
__global__ __launch_bounds__(1024) void k6_select(const uint8_t* __restrict__ status, const float* __restrict__ pcs,
                                                  const float* __restrict__ rcs, int n_tracks, int min_new_points,
                                                  int32_t* __restrict__ accepted, int32_t* __restrict__ inconsistent,
                                                  int32_t* __restrict__ rejected, int32_t* __restrict__ counts)
{
    const int T = blockDim.x, chunk = (n_tracks + T - 1) / T;
    const int lo = min((int)threadIdx.x * chunk, n_tracks), hi = min(lo + chunk, n_tracks);
    int na = 0, nr = 0, ni = 0;
    for (int t = lo; t < hi; t++) {
        const int st = status[t];
        if (st == 2) ni++;
        else if (st == 1) { if (pcs[t] <= rcs[t]) na++; else nr++; }
    }
    int tot_a, tot_r, tot_i;
    int oa = rs_block_exclusive_scan(na, &tot_a);
    int orj = rs_block_exclusive_scan(nr, &tot_r);
    int oi = rs_block_exclusive_scan(ni, &tot_i);
    for (int t = lo; t < hi; t++) {
        const int st = status[t];
        if (st == 2) inconsistent[oi++] = t;
        else if (st == 1) { if (pcs[t] <= rcs[t]) accepted[oa++] = t; else rejected[orj++] = t; }
    }
    __threadfence_block();
    __syncthreads();
    int top = 0;
    if (tot_a < min_new_points && tot_r > 0) {
        top = min(min_new_points - tot_a, tot_r);
        for (int i = threadIdx.x; i < tot_r; i += T) {
            const int ti = rejected[i];
            const float ki = pcs[ti];
            int rank = 0;
            for (int j = 0; j < tot_r; j++) {
                const float kj = pcs[rejected[j]];
                // kj sorts before ki:  kj < ki, or equal (or both unordered) and earlier; NaN after everything
                const bool before = (kj < ki) || (!(ki < kj) && !(kj < ki) && ((kj == kj) == (ki == ki)) && j < i) ||
                                    ((kj == kj) && !(ki == ki));
                rank += before ? 1 : 0;
            }
            if (rank < top) accepted[tot_a + rank] = ti;
        }
    }
    if (threadIdx.x == 0) { counts[0] = tot_a + top; counts[1] = top; counts[2] = tot_i; }
}

extern "C" int rs_triangulate_tracks(rs_context* ctx, int n_tracks, const float* d_track_uv, const uint8_t* d_skip,
                                     const int32_t* d_sight_ptr, const int32_t* d_sight_pose,
                                     const float* d_sight_uv, const float* d_poses, int n_poses, int kf_pose,
                                     const float h_intrinsics[4], float any_parallax_cosine,
                                     float max_reprojection_error, float min_parallax_cosine,
                                     float rotation_parallax_factor, int min_new_points, uint8_t* d_status,
                                     float* d_xyz, float* d_parallax_cos, float* d_required_cos,
                                     int32_t* d_accepted, int32_t* d_inconsistent, int32_t* d_counts,
                                     const float* d_required_by_pose)
{
    if (!ctx) return RS_ERR_INVALID;
    if (n_tracks < 0 || n_poses < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (!d_counts) return rs_fail(ctx, RS_ERR_INVALID, "null counts");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    if (n_tracks == 0) {
        RS_HIP(ctx, hipMemsetAsync(d_counts, 0, 3 * sizeof(int32_t), ctx->stream));
        return RS_OK;
    }
    if (!d_track_uv || !d_sight_ptr || !d_sight_pose || !d_sight_uv || !d_poses || !h_intrinsics || !d_status ||
        !d_xyz || !d_parallax_cos || !d_required_cos || !d_accepted || !d_inconsistent)
        return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    if (kf_pose < 0 || kf_pose >= n_poses) return rs_fail(ctx, RS_ERR_INVALID, "key-frame pose index out of range");
    if ((uintptr_t)d_poses & 15) return rs_fail(ctx, RS_ERR_INVALID, "poses must be 16-byte aligned");
    void* ws = nullptr;
    int rc = rs_workspace(ctx, sizeof(int32_t) * (size_t)n_tracks, &ws);
    if (rc) return rc;
    TrackParams prm;
    prm.tri = {h_intrinsics[0], h_intrinsics[1], h_intrinsics[2], h_intrinsics[3], any_parallax_cosine, max_reprojection_error};
    prm.min_parallax_cosine = min_parallax_cosine;
    prm.rotation_factor = rotation_parallax_factor;
    prm.kf_pose = kf_pose;
    {
        rs_prof_scope ps(ctx, "K6_tracks");
        hipLaunchKernelGGL(k6_tracks, dim3((n_tracks + 63) / 64), dim3(64), 0, ctx->stream, (const float2*)d_track_uv,
                           d_skip, d_sight_ptr, d_sight_pose, (const float2*)d_sight_uv, d_poses, n_tracks, prm,
                           d_status, d_xyz, d_parallax_cos, d_required_cos, d_required_by_pose);
    }
    {
        rs_prof_scope ps(ctx, "K6b_select");
        hipLaunchKernelGGL(k6_select, dim3(1), dim3(1024), 0, ctx->stream, d_status, d_parallax_cos, d_required_cos,
                           n_tracks, min_new_points, d_accepted, d_inconsistent, (int32_t*)ws, d_counts);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}

// ---------------------------------------------------------------------------------------------------
// K12: per-point mean reprojection error and the culling rule of Mapper::cull_points (reference
// src/Mapper.cpp:396-431), plus the sums Slam::reprojection_error needs
// (src/Slam.cpp:302-317).  One lane per point, observations in CSR order; the per-point error is the f32 sum
// of (project(pose, X) - pixel).norm() in that order (the reference iterates an unordered_map: order
// unspecified upstream), divided by the count; cull when mean > max_mean_error and the point has observations.
__global__ __launch_bounds__(256) void k12_point_errors(const float* __restrict__ pos, const int32_t* __restrict__ obs_ptr,
                                                        const int32_t* __restrict__ obs_pose,
                                                        const float2* __restrict__ obs_uv, const float* __restrict__ poses,
                                                        int n_points, TriParams k, float max_mean_error,
                                                        float* __restrict__ mean_err, uint8_t* __restrict__ cull,
                                                        double* __restrict__ sums)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double esum = 0.0, ecnt = 0.0;
    if (p < n_points) {
        const float X[3] = {pos[3 * (size_t)p], pos[3 * (size_t)p + 1], pos[3 * (size_t)p + 2]};
        const int o0 = obs_ptr[p], o1 = obs_ptr[p + 1];
        float err = 0.0f;
        for (int o = o0; o < o1; o++) {
            float T[16];
            load_pose(poses, obs_pose[o], T);
            const float2 pr = project_f32(k, T, X);
            const float2 px = obs_uv[o];
            const float dx = pr.x - px.x, dy = pr.y - px.y;
            const float e = sqrtf(dx * dx + dy * dy);
            err += e;                                                    // :413
            esum += (double)e;
        }
        const int cnt = o1 - o0;
        ecnt = (double)cnt;
        const float mean = cnt > 0 ? err / (float)cnt : 0.0f;
        mean_err[p] = mean;
        cull[p] = (cnt > 0 && mean > max_mean_error) ? 1 : 0;            // :416
    }
    // Slam::reprojection_error: sum and count over all observations (f64 sum of the f32 errors)
    esum = wave_sum_f64(esum);
    ecnt = wave_sum_f64(ecnt);
    if ((threadIdx.x & 63) == 0 && ecnt > 0.0) { atomicAdd(&sums[0], esum); atomicAdd(&sums[1], ecnt); }
}

// ordered list of the culled points (one workgroup)
__global__ __launch_bounds__(1024) void k12_cull_list(const uint8_t* __restrict__ cull, int n, int32_t* __restrict__ out_idx,
                                                      int32_t* __restrict__ out_count)
{
    const int T = blockDim.x, chunk = (n + T - 1) / T;
    const int lo = min((int)threadIdx.x * chunk, n), hi = min(lo + chunk, n);
    int cnt = 0;
    for (int i0 = lo; i0 < hi; i0 += 16) {
        uint8_t k[16];
#pragma unroll
        for (int u = 0; u < 16; u++) k[u] = cull[min(i0 + u, hi - 1)];
#pragma unroll
        for (int u = 0; u < 16; u++) cnt += (i0 + u < hi && k[u] != 0) ? 1 : 0;
    }
    int total;
    int off = rs_block_exclusive_scan(cnt, &total);
    for (int i = lo; i < hi; i++)
        if (cull[i]) out_idx[off++] = i;
    if (threadIdx.x == 0) *out_count = total;
}

extern "C" int rs_point_errors(rs_context* ctx, int n_points, const float* d_positions, const int32_t* d_obs_ptr,
                               const int32_t* d_obs_pose, const float* d_obs_uv, const float* d_poses, int n_poses,
                               const float h_intrinsics[4], float max_mean_error, float* d_mean_err, uint8_t* d_cull,
                               int32_t* d_cull_idx, int32_t* d_cull_count, double* d_sums)
{
    if (!ctx) return RS_ERR_INVALID;
    if (n_points < 0 || n_poses < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (!d_cull_count || !d_sums) return rs_fail(ctx, RS_ERR_INVALID, "null output");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    RS_HIP(ctx, hipMemsetAsync(d_sums, 0, 2 * sizeof(double), ctx->stream));
    if (n_points == 0) {
        RS_HIP(ctx, hipMemsetAsync(d_cull_count, 0, sizeof(int32_t), ctx->stream));
        return RS_OK;
    }
    if (!d_positions || !d_obs_ptr || !d_obs_pose || !d_obs_uv || !d_poses || !h_intrinsics || !d_mean_err || !d_cull || !d_cull_idx)
        return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    if ((uintptr_t)d_poses & 15) return rs_fail(ctx, RS_ERR_INVALID, "poses must be 16-byte aligned");
    const TriParams k = {h_intrinsics[0], h_intrinsics[1], h_intrinsics[2], h_intrinsics[3], 0.f, 0.f};
    {
        rs_prof_scope ps(ctx, "K12_point_errors");
        hipLaunchKernelGGL(k12_point_errors, dim3((n_points + 255) / 256), dim3(256), 0, ctx->stream, d_positions, d_obs_ptr,
                           d_obs_pose, (const float2*)d_obs_uv, d_poses, n_points, k, max_mean_error, d_mean_err, d_cull, d_sums);
    }
    {
        rs_prof_scope ps(ctx, "K12b_cull_list");
        hipLaunchKernelGGL(k12_cull_list, dim3(1), dim3(1024), 0, ctx->stream, d_cull, n_points, d_cull_idx, d_cull_count);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}

// ---------------------------------------------------------------------- K13
// The tail of Mapper::bundle_adjust (reference src/Mapper.cpp:380-393): points with a single observation were not
// optimised; they are carried along rigidly with the frame that observes them,
//     in_camera = R_before X + t_before;   X' = R_after^T (in_camera - t_after)          (all f32)
// One lane per listed point.  f32 operation order as everywhere in this library: a 3-term dot product is
// (a0 b0 + a1 b1) + a2 b2 (Eigen's own order is unspecified upstream).
__global__ __launch_bounds__(256) void k13_reanchor(int n, const int32_t* __restrict__ point_idx,
                                                    const int32_t* __restrict__ frame_idx, int n_frames,
                                                    const float* __restrict__ before, const float* __restrict__ after,
                                                    float* __restrict__ pos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = point_idx ? point_idx[i] : i;
    if ((unsigned)frame_idx[i] >= (unsigned)n_frames) return;      // no such frame: the point stays where it is (both K13 forms)
    float B[16], A[16];
    load_pose(before, frame_idx[i], B);
    load_pose(after, frame_idx[i], A);
    const float X[3] = {pos[3 * (size_t)p], pos[3 * (size_t)p + 1], pos[3 * (size_t)p + 2]};
    float c[3], d[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        c[r] = ((B[4 * r] * X[0] + B[4 * r + 1] * X[1]) + B[4 * r + 2] * X[2]) + B[4 * r + 3];     // :389
        d[r] = c[r] - A[4 * r + 3];
    }
#pragma unroll
    for (int r = 0; r < 3; r++)                                                                       // :390
        pos[3 * (size_t)p + r] = (A[r] * d[0] + A[4 + r] * d[1]) + A[8 + r] * d[2];
}

extern "C" int rs_reanchor_points(rs_context* ctx, int n, const int32_t* d_point_idx, const int32_t* d_frame_idx,
                                  const float* d_poses_before, const float* d_poses_after, int n_frames,
                                  float* d_positions)
{
    if (!ctx) return RS_ERR_INVALID;
    if (n < 0 || n_frames < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (n == 0) return RS_OK;
    if (!d_frame_idx || !d_poses_before || !d_poses_after || !d_positions) return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    if (((uintptr_t)d_poses_before | (uintptr_t)d_poses_after) & 15) return rs_fail(ctx, RS_ERR_INVALID, "poses must be 16-byte aligned");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    {
        rs_prof_scope ps(ctx, "K13_reanchor");
        hipLaunchKernelGGL(k13_reanchor, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_point_idx, d_frame_idx, n_frames,
                           d_poses_before, d_poses_after, d_positions);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}

// K13 with the poses in the kernel's argument block: rows 0..2 of each 4 x 4 pose (the arithmetic reads nothing else)
#define K13_ARG_FRAMES 32
struct K13Poses { float before[K13_ARG_FRAMES][12]; float after[K13_ARG_FRAMES][12]; };
__global__ __launch_bounds__(256) void k13_reanchor_args(int n, const int32_t* __restrict__ point_idx,
                                                         const int32_t* __restrict__ frame_idx, int n_frames, K13Poses poses,
                                                         float* __restrict__ pos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = point_idx ? point_idx[i] : i;
    const int f = frame_idx[i];
    if ((unsigned)f >= (unsigned)n_frames) return;                 // (an index beyond the argument block would read past it)
    const float* B = poses.before[f];
    const float* A = poses.after[f];
    const float X[3] = {pos[3 * (size_t)p], pos[3 * (size_t)p + 1], pos[3 * (size_t)p + 2]};
    float c[3], d[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        c[r] = ((B[4 * r] * X[0] + B[4 * r + 1] * X[1]) + B[4 * r + 2] * X[2]) + B[4 * r + 3];     // :389
        d[r] = c[r] - A[4 * r + 3];
    }
#pragma unroll
    for (int r = 0; r < 3; r++)                                                                       // :390
        pos[3 * (size_t)p + r] = (A[r] * d[0] + A[4 + r] * d[1]) + A[8 + r] * d[2];
}

extern "C" int rs_reanchor_points_host_poses(rs_context* ctx, int n, const int32_t* d_point_idx, const int32_t* d_frame_idx,
                                             const float* h_poses_before, const float* h_poses_after, int n_frames,
                                             float* d_positions)
{
    if (!ctx) return RS_ERR_INVALID;
    if (n < 0 || n_frames < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (n == 0) return RS_OK;
    if (!d_frame_idx || !h_poses_before || !h_poses_after || !d_positions || n_frames == 0) return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    if (n_frames <= K13_ARG_FRAMES) {
        K13Poses P;
        for (int f = 0; f < n_frames; f++) {
            memcpy(P.before[f], h_poses_before + 16 * (size_t)f, sizeof(float) * 12);
            memcpy(P.after[f], h_poses_after + 16 * (size_t)f, sizeof(float) * 12);
        }
        for (int f = n_frames; f < K13_ARG_FRAMES; f++) { memset(P.before[f], 0, sizeof P.before[f]); memset(P.after[f], 0, sizeof P.after[f]); }
        rs_prof_scope ps(ctx, "K13_reanchor");
        hipLaunchKernelGGL(k13_reanchor_args, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_point_idx, d_frame_idx, n_frames, P, d_positions);
    } else {
        // larger sets of frames: both pose arrays through the context's workspace (copied before this call returns)
        void* ws = nullptr;
        const size_t bytes = sizeof(float) * 16 * (size_t)n_frames;
        const int rc = rs_workspace(ctx, 2 * bytes + 256, &ws);
        if (rc) return rc;
        float* d_before = (float*)ws;
        float* d_after = (float*)((char*)ws + ((bytes + 255) & ~(size_t)255));
        RS_HIP(ctx, hipMemcpyAsync(d_before, h_poses_before, bytes, hipMemcpyHostToDevice, ctx->stream));
        RS_HIP(ctx, hipMemcpyAsync(d_after, h_poses_after, bytes, hipMemcpyHostToDevice, ctx->stream));
        RS_HIP(ctx, hipStreamSynchronize(ctx->stream));          // pageable sources: the copies have read them
        rs_prof_scope ps(ctx, "K13_reanchor");
        hipLaunchKernelGGL(k13_reanchor, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_point_idx, d_frame_idx, n_frames,
                           d_before, d_after, d_positions);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}

// ------------------------------------------------------------------------------------------------ K14
// transform_points of the pose graph (reference src/Optimization.cpp:512-536): after a loop closure moved the key
// frames, every map point that has observations moves rigidly with its OWNER, the observing key frame of smallest
// index — same arithmetic as K13, one lane per point over the whole map, the owner found by a scan of the point's
// (short) observation list.  obs_kf holds positions in the key-frame list handed to pose_graph, which is in index
// order (Mapper::key_frames(), src/Slam.cpp:263); a point whose owner is not in [0, n_kf) stays (:524-527).
// HBM-bound: 8 B of obs_ptr + 4 B per observation + 24 B of position per point; the two 4x4 poses come from L2.
__global__ __launch_bounds__(256) void k14_transform_points(int n_points, const int32_t* __restrict__ obs_ptr,
                                                            const int32_t* __restrict__ obs_kf, int n_kf,
                                                            const float* __restrict__ before, const float* __restrict__ after,
                                                            float* __restrict__ pos)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_points) return;
    const int lo = obs_ptr[p], hi = obs_ptr[p + 1];
    if (hi <= lo) return;                                                                             // :515-517
    int owner = obs_kf[lo];
    for (int o = lo + 1; o < hi; o++) owner = min(owner, obs_kf[o]);                                  // :518-523
    if (owner < 0 || owner >= n_kf) return;
    float B[16], A[16];
    load_pose(before, owner, B);
    load_pose(after, owner, A);
    const float X[3] = {pos[3 * (size_t)p], pos[3 * (size_t)p + 1], pos[3 * (size_t)p + 2]};
    float c[3], d[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        c[r] = ((B[4 * r] * X[0] + B[4 * r + 1] * X[1]) + B[4 * r + 2] * X[2]) + B[4 * r + 3];     // :531
        d[r] = c[r] - A[4 * r + 3];
    }
#pragma unroll
    for (int r = 0; r < 3; r++)                                                                       // :532
        pos[3 * (size_t)p + r] = (A[r] * d[0] + A[4 + r] * d[1]) + A[8 + r] * d[2];
}

extern "C" int rs_transform_points(rs_context* ctx, int n_points, const int32_t* d_obs_ptr, const int32_t* d_obs_kf,
                                   const float* d_poses_before, const float* d_poses_after, int n_kf, float* d_positions)
{
    if (!ctx) return RS_ERR_INVALID;
    if (n_points < 0 || n_kf < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (n_points == 0 || n_kf == 0) return RS_OK;
    if (!d_obs_ptr || !d_obs_kf || !d_poses_before || !d_poses_after || !d_positions) return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    if (((uintptr_t)d_poses_before | (uintptr_t)d_poses_after) & 15) return rs_fail(ctx, RS_ERR_INVALID, "poses must be 16-byte aligned");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    {
        rs_prof_scope ps(ctx, "K14_transform_points");
        hipLaunchKernelGGL(k14_transform_points, dim3((n_points + 255) / 256), dim3(256), 0, ctx->stream, n_points, d_obs_ptr,
                           d_obs_kf, n_kf, d_poses_before, d_poses_after, d_positions);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}
