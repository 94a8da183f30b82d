/* tracks.c — CPU restatement of the body of Mapper::triangulate_tracks
 * (reference src/Mapper.cpp:246-305).
 *
 * TEST INFRASTRUCTURE ONLY (see rs_oracle.h).  PARITY UNPINNED: the reference has no
 * tests or fixtures for this path and cannot be built here.
 *
 * Per track, in track-id order (the reference iterates a std::map<TrackId, Track>,
 * src/TrackStore.h:38): triangulate (first sighting, key-frame pixel) with the loose gates
 * (cos <= 1.0, 4 px), src/Mapper.cpp:252-262; reproject into EVERY sighting's pose and drop
 * the track as inconsistent on the first error > 4 px, :264-275; parallax cosine between the
 * rays to the first sighting's camera and to the key frame, and the rotation-dependent
 * requirement min(cos 1 deg, cos(0.2 * turned)), :277-288.  Then the selection :291-304:
 * every candidate at or below its requirement, in order; if fewer than the quota (100), the
 * rest sorted by parallax cosine ascending tops it up.  The reference sorts with std::sort
 * (order of equal cosines unspecified); here ties keep candidate order.
 * The host-side filters of :247-250 (key point already matched, empty track) arrive as `skip`.
 * float math: Eigen expressions restated operation by operation as in triangulate.c /
 * reproj_match.c; acosf / cosf are libm's (the GPU's differ in the last ulp: DESIGN.md §2). */
#include <math.h>
#include <stdlib.h>

#include "rs_oracle.h"

static float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

static void normalize3(float* v)
{
    const float n = dot3(v, v);
    if (n > 0.0f) {
        const float s = sqrtf(n);
        v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s;
    }
}

/* Camera::project, src/Camera.cpp:25-32 */
static void project(const float K[4], const float* T, const float* X, float uv[2])
{
    float KP[12];
    for (int j = 0; j < 4; j++) {
        KP[0 * 4 + j] = (K[0] * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + K[2] * T[2 * 4 + j];
        KP[1 * 4 + j] = (0.0f * T[0 * 4 + j] + K[1] * T[1 * 4 + j]) + K[3] * T[2 * 4 + j];
        KP[2 * 4 + j] = (0.0f * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + 1.0f * T[2 * 4 + j];
    }
    float uvw[3];
    for (int i = 0; i < 3; i++) {
        const float* r = KP + 4 * i;
        uvw[i] = (r[0] * X[0] + r[1] * X[1]) + (r[2] * X[2] + r[3] * 1.0f);
    }
    if (uvw[2] < 0.0f) { uv[0] = -1.0f; uv[1] = -1.0f; }
    else { uv[0] = uvw[0] / uvw[2]; uv[1] = uvw[1] / uvw[2]; }
}

/* motion::camera_center / Frame::camera_center = -R^T t, src/MotionModel.cpp:8-11, src/Frame.cpp:39-42 */
static void camera_center(const float* T, float c[3])
{
    const float t[3] = {T[3], T[7], T[11]};
    for (int i = 0; i < 3; i++) {
        const float a[3] = {-T[0 * 4 + i], -T[1 * 4 + i], -T[2 * 4 + i]};
        c[i] = dot3(a, t);
    }
}

int orc_triangulate_tracks(int n_tracks, const float* track_uv, const uint8_t* skip, const int32_t* sight_ptr,
                           const int32_t* sight_pose, const float* sight_uv, const float* poses, int n_poses,
                           int kf_pose, const float K[4], float any_parallax_cosine,
                           float max_reprojection_error, float min_parallax_cosine,
                           float rotation_parallax_factor, int min_new_points, uint8_t* status, float* xyz,
                           float* parallax_cos, float* required_cos, int32_t* accepted, int32_t* n_accepted,
                           int32_t* n_topped_up, int32_t* inconsistent, int32_t* n_inconsistent)
{
    *n_accepted = 0; *n_topped_up = 0; *n_inconsistent = 0;
    if (n_tracks <= 0) return 0;
    if (kf_pose < 0 || kf_pose >= n_poses) return 1;
    const size_t T = (size_t)n_tracks;
    float* uv1 = (float*)malloc(sizeof(float) * 2 * T);
    int32_t* idx1 = (int32_t*)malloc(sizeof(int32_t) * T);
    int32_t* idx2 = (int32_t*)malloc(sizeof(int32_t) * T);
    uint8_t* keep = (uint8_t*)malloc(T);
    int32_t* oi = (int32_t*)malloc(sizeof(int32_t) * T);
    float* ox = (float*)malloc(sizeof(float) * 3 * T);
    int32_t cnt = 0;
    for (size_t t = 0; t < T; t++) {
        const int s0 = sight_ptr[t], s1 = sight_ptr[t + 1];
        const int has = s1 > s0 && !(skip && skip[t]);
        uv1[2 * t] = has ? sight_uv[2 * (size_t)s0] : 0.0f;
        uv1[2 * t + 1] = has ? sight_uv[2 * (size_t)s0 + 1] : 0.0f;
        idx1[t] = has ? sight_pose[s0] : kf_pose;
        idx2[t] = kf_pose;
    }
    /* triangulation::triangulate_points on one correspondence per track, src/Mapper.cpp:254-260 */
    int rc = orc_triangulate(uv1, track_uv, n_tracks, poses, n_poses, idx1, idx2, K, any_parallax_cosine,
                             max_reprojection_error, xyz, keep, oi, ox, &cnt);
    if (rc) goto done;
    const float* Tk = poses + 16 * (size_t)kf_pose;
    float ck[3];
    camera_center(Tk, ck);
    int ninc = 0;
    for (size_t t = 0; t < T; t++) {
        status[t] = 0; parallax_cos[t] = 0.0f; required_cos[t] = 0.0f;
        const int s0 = sight_ptr[t], s1 = sight_ptr[t + 1];
        if (s1 <= s0 || (skip && skip[t])) {                            /* :248-250: never triangulated */
            xyz[3 * t] = 0.0f; xyz[3 * t + 1] = 0.0f; xyz[3 * t + 2] = 0.0f;
            continue;
        }
        if (!keep[t]) continue;                                          /* :261-263 */
        const float* X = xyz + 3 * t;
        int consistent = 1;
        for (int s = s0; s < s1; s++) {                                   /* :265-271 */
            float pr[2];
            project(K, poses + 16 * (size_t)sight_pose[s], X, pr);
            const float dx = pr[0] - sight_uv[2 * (size_t)s], dy = pr[1] - sight_uv[2 * (size_t)s + 1];
            if (sqrtf(dx * dx + dy * dy) > max_reprojection_error) { consistent = 0; break; }
        }
        if (!consistent) { status[t] = 2; inconsistent[ninc++] = (int32_t)t; continue; }   /* :272-275 */
        const float* Tf = poses + 16 * (size_t)sight_pose[s0];
        float cf[3];
        camera_center(Tf, cf);
        float a[3] = {cf[0] - X[0], cf[1] - X[1], cf[2] - X[2]};
        float b[3] = {ck[0] - X[0], ck[1] - X[1], ck[2] - X[2]};
        normalize3(a);                                                    /* :278-279 */
        normalize3(b);
        /* turn = R_kf * R_first^T, :281-282 */
        float tr[3];
        for (int i = 0; i < 3; i++) {
            const float rk[3] = {Tk[4 * i], Tk[4 * i + 1], Tk[4 * i + 2]};
            const float rf[3] = {Tf[4 * i], Tf[4 * i + 1], Tf[4 * i + 2]};
            tr[i] = dot3(rk, rf);
        }
        const float trace = (tr[0] + tr[1]) + tr[2];
        float cosine = (trace - 1.0f) / 2.0f;
        cosine = cosine < -1.0f ? -1.0f : cosine;
        cosine = cosine > 1.0f ? 1.0f : cosine;
        const float turned = acosf(cosine);
        const float need = cosf(rotation_parallax_factor * turned);
        status[t] = 1;
        parallax_cos[t] = dot3(a, b);                                     /* :287 */
        required_cos[t] = min_parallax_cosine < need ? min_parallax_cosine : need;   /* :288 */
    }
    *n_inconsistent = ninc;
    /* selection, :291-304 */
    {
        int na = 0, nr = 0;
        int32_t* rej = (int32_t*)malloc(sizeof(int32_t) * T);
        for (size_t t = 0; t < T; t++) {
            if (status[t] != 1) continue;
            if (parallax_cos[t] <= required_cos[t]) accepted[na++] = (int32_t)t; else rej[nr++] = (int32_t)t;
        }
        int top = 0;
        if (na < min_new_points && nr > 0) {
            /* stable insertion sort by parallax cosine ascending (ties keep candidate order) */
            for (int i = 1; i < nr; i++) {
                const int32_t v = rej[i];
                int j = i - 1;
                while (j >= 0 && parallax_cos[rej[j]] > parallax_cos[v]) { rej[j + 1] = rej[j]; j--; }
                rej[j + 1] = v;
            }
            top = min_new_points - na < nr ? min_new_points - na : nr;
            for (int i = 0; i < top; i++) accepted[na + i] = rej[i];
        }
        *n_accepted = na + top;
        *n_topped_up = top;
        free(rej);
    }
done:
    free(uv1); free(idx1); free(idx2); free(keep); free(oi); free(ox);
    return rc;
}

/* Mapper::cull_points arithmetic (src/Mapper.cpp:410-419) and the sums of Slam::reprojection_error
 * (src/Slam.cpp:302-317).  Observations of a point are visited in CSR order (the reference iterates an
 * unordered_map: order unspecified upstream); f32 accumulation per point, f64 for the global sum. */
int orc_point_errors(int n_points, const float* positions, const int32_t* obs_ptr, const int32_t* obs_pose,
                     const float* obs_uv, const float* poses, int n_poses, const float K[4], float max_mean_error,
                     float* mean_err, uint8_t* cull, int32_t* cull_idx, int32_t* cull_count, double sums[2])
{
    (void)n_poses;
    int nc = 0;
    sums[0] = 0.0; sums[1] = 0.0;
    for (int p = 0; p < n_points; p++) {
        const float* X = positions + 3 * (size_t)p;
        float error = 0.0f;
        int num = 0;
        for (int o = obs_ptr[p]; o < obs_ptr[p + 1]; o++) {
            float pr[2];
            project(K, poses + 16 * (size_t)obs_pose[o], X, pr);
            const float dx = pr[0] - obs_uv[2 * (size_t)o], dy = pr[1] - obs_uv[2 * (size_t)o + 1];
            const float e = sqrtf(dx * dx + dy * dy);
            error += e;                                                 /* :413 */
            sums[0] += (double)e;
            num++;
        }
        sums[1] += (double)num;
        mean_err[p] = num > 0 ? error / (float)num : 0.0f;
        cull[p] = (num > 0 && error / (float)num > max_mean_error) ? 1 : 0;   /* :416 */
        if (cull[p]) cull_idx[nc++] = p;
    }
    *cull_count = nc;
    return 0;
}

/* The tail of Mapper::bundle_adjust, reference src/Mapper.cpp:380-393: a single-observation point keeps its place
 * relative to the frame that observes it.  f32; 3-term dot products as (a0 b0 + a1 b1) + a2 b2. */
int orc_reanchor_points(int n, const int32_t* point_idx, const int32_t* frame_idx, const float* before,
                        const float* after, float* positions)
{
    for (int i = 0; i < n; i++) {
        const int p = point_idx ? point_idx[i] : i;
        const float* B = before + 16 * (size_t)frame_idx[i];
        const float* A = after + 16 * (size_t)frame_idx[i];
        float* X = positions + 3 * (size_t)p;
        float c[3], d[3];
        for (int r = 0; r < 3; r++) {
            c[r] = ((B[4 * r] * X[0] + B[4 * r + 1] * X[1]) + B[4 * r + 2] * X[2]) + B[4 * r + 3];   /* :389 */
            d[r] = c[r] - A[4 * r + 3];
        }
        for (int r = 0; r < 3; r++) X[r] = (A[r] * d[0] + A[4 + r] * d[1]) + A[8 + r] * d[2];        /* :390 */
    }
    return 0;
}

/* transform_points, reference src/Optimization.cpp:512-536 (the point loop of optimization::pose_graph). */
int orc_transform_points(int n_points, const int32_t* obs_ptr, const int32_t* obs_kf, const float* before,
                         const float* after, float* positions)
{
    for (int p = 0; p < n_points; p++) {
        if (obs_ptr[p + 1] == obs_ptr[p]) continue;                                  /* :515-517 */
        int owner = obs_kf[obs_ptr[p]];
        for (int o = obs_ptr[p] + 1; o < obs_ptr[p + 1]; o++)
            if (obs_kf[o] < owner) owner = obs_kf[o];                                /* :518-523 */
        if (owner < 0) continue;                                                     /* :524-527 */
        const int32_t one = owner;
        orc_reanchor_points(1, &p, &one, before, after, positions);                  /* :528-534, same arithmetic as Mapper.cpp:389-390 */
    }
    return 0;
}
