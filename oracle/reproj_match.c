/*
 * oracle/reproj_match.c — CPU restatement of MapMatcher::match /
 * match_for_fuse (reference src/MapMatcher.cpp:25-98,107-127,165-175).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see rs_oracle.h).
 *
 * Specified f32 operation order (the reference leaves it to Eigen's
 * expression templates; no FMA contraction: the reference builds for baseline
 * x86-64, CMakeLists.txt sets no -march):
 *   dot3(a,b)  = (a0*b0 + a1*b1) + a2*b2
 *   dot4(a,b)  = (a0*b0 + a1*b1) + (a2*b2 + a3*b3)
 *   normalized = v / sqrt(dot3(v,v))   (unchanged when the squared norm is 0)
 * Observation order = CSR order (the reference iterates an unordered_map:
 * unspecified and not run-to-run stable, SURVEY.md §7 "bit-exactness hazards").
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <string.h>

#include "rs_oracle.h"

#define SEARCH_RADIUS 20.0f            /* src/MapMatcher.cpp:12 */
#define MIN_VIEWING_ANGLE_COSINE 0.5f  /* :13 */
#define MAX_NEARER_RATIO 2.0f          /* :16 */
#define MAX_FURTHER_RATIO 1.25f        /* :17 */

static float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

static void normalize3(float* v)
{
    float n = dot3(v, v);
    if (n > 0.0f) {
        float s = sqrtf(n);
        v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s;
    }
}

static int hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

int orc_reproj_match(const orc_frame_view* f, const orc_map_view* m, int replace,
                     int max_distance, int32_t* point_kp, int32_t* point_dist,
                     int32_t* prop_point, int32_t* prop_dist, int32_t* match_kp,
                     int32_t* match_point, int32_t* match_count)
{
    const int N = f->n_keypoints, P = m->n_points;
    const float* T = f->pose;
    /* empty_proposals, src/MapMatcher.cpp:25-32 */
    for (int i = 0; i < N; i++) { prop_point[i] = -1; prop_dist[i] = max_distance; }

    /* Frame::camera_center = -R^T t, src/Frame.cpp:39-42 */
    float center[3];
    for (int i = 0; i < 3; i++) {
        float a[3] = {-T[0 * 4 + i], -T[1 * 4 + i], -T[2 * 4 + i]};
        float t[3] = {T[3], T[7], T[11]};
        center[i] = dot3(a, t);
    }
    /* K * pose.block<3,4>, src/Camera.cpp:27: K has zeros off (0,0),(1,1),(0,2),(1,2),(2,2) */
    float KP[12];
    for (int j = 0; j < 4; j++) {
        KP[0 * 4 + j] = (f->fx * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + f->cx * T[2 * 4 + j];
        KP[1 * 4 + j] = (0.0f * T[0 * 4 + j] + f->fy * T[1 * 4 + j]) + f->cy * T[2 * 4 + j];
        KP[2 * 4 + j] = (0.0f * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + 1.0f * T[2 * 4 + j];
    }

    /* Per-point work first (independent: the all-cores baseline build runs it in parallel), then the per-keypoint
     * proposal table in MAP ORDER (:95-97), which is what makes the result order dependent. */
    int32_t* raw_kp = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(P > 0 ? P : 1));
    int32_t* raw_d = raw_kp + (P > 0 ? P : 1);
#ifdef ORC_OMP
#pragma omp parallel
#endif
    {
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
#ifdef ORC_OMP
#pragma omp for schedule(static, 64)
#endif
    for (int p = 0; p < P; p++) {
        point_kp[p] = -1;
        point_dist[p] = max_distance;
        raw_kp[p] = -1;
        if (!m->eligible[p]) continue;            /* :53, :169, :121-123 folded by the caller */
        const float* X = m->positions + 3 * (size_t)p;
        /* Camera::project, src/Camera.cpp:25-32 */
        float uvw[3];
        for (int i = 0; i < 3; i++) {
            const float* r = KP + 4 * i;
            uvw[i] = (r[0] * X[0] + r[1] * X[1]) + (r[2] * X[2] + r[3] * 1.0f);
        }
        float u, v;
        if (uvw[2] < 0.0f) { u = -1.0f; v = -1.0f; }
        else { u = uvw[0] / uvw[2]; v = uvw[1] / uvw[2]; }
        /* is_in_image, src/Camera.cpp:34-37 */
        if (!(u >= 0.0f && u < (float)f->width && v >= 0.0f && v < (float)f->height)) continue;

        float ray[3] = {X[0] - center[0], X[1] - center[1], X[2] - center[2]};
        /* avg_viewing_normal + observed_distance_range, src/MapPoint.cpp:24-45 */
        float normal[3] = {0.0f, 0.0f, 0.0f};
        float nearest = FLT_MAX, furthest = 0.0f;
        for (int o = m->obs_ptr[p]; o < m->obs_ptr[p + 1]; o++) {
            const float* C = m->kf_centers + 3 * (size_t)m->obs_kf[o];
            float d[3] = {X[0] - C[0], X[1] - C[1], X[2] - C[2]};
            float dist = sqrtf(dot3(d, d));
            nearest = dist < nearest ? dist : nearest;
            furthest = furthest < dist ? dist : furthest;
            normalize3(d);
            normalize3(d);     /* v.normalized() of an already normalized v, :29-30 */
            normal[0] += d[0]; normal[1] += d[1]; normal[2] += d[2];
        }
        normalize3(normal);
        float rn[3] = {ray[0], ray[1], ray[2]};
        normalize3(rn);
        if (dot3(normal, rn) < MIN_VIEWING_ANGLE_COSINE) continue;          /* :62-66 */
        float distance = sqrtf(dot3(ray, ray));
        if (distance < nearest / MAX_NEARER_RATIO || distance > furthest * MAX_FURTHER_RATIO)
            continue;                                                       /* :69-73 */

        int nc = orc_kdtree_radius(f->keypoints, f->kd_node_kp, f->kd_left, f->kd_right,
                                   f->kd_root, u, v, SEARCH_RADIUS, cand, N);   /* :75 */
        int best_kp = 0, best_d = max_distance;                             /* :77-78 */
        for (int c = 0; c < nc; c++) {
            int kp = cand[c];
            if (!replace && f->kp_matched[kp]) continue;                    /* :81 */
            const uint8_t* d = f->descriptors + 32 * (size_t)kp;
            for (int o = m->obs_ptr[p]; o < m->obs_ptr[p + 1]; o++) {
                int hd = hamming256(d, m->desc_pool + 32 * (size_t)m->obs_desc[o]);
                if (hd < best_d) { best_d = hd; best_kp = kp; }             /* :88-91 */
            }
        }
        if (best_d < max_distance) { point_kp[p] = best_kp; point_dist[p] = best_d; }
        raw_kp[p] = best_kp; raw_d[p] = best_d;
    }
    free(cand);
    }
    for (int p = 0; p < P; p++) {
        if (raw_kp[p] < 0) continue;
        if (raw_d[p] < prop_dist[raw_kp[p]]) {                              /* :95-97 */
            prop_point[raw_kp[p]] = p;
            prop_dist[raw_kp[p]] = raw_d[p];
        }
    }
    free(raw_kp);
    /* accepted_matches, :34-43 */
    int count = 0;
    for (int i = 0; i < N; i++) {
        if (prop_point[i] >= 0) { match_kp[count] = i; match_point[count] = prop_point[i]; count++; }
    }
    *match_count = count;
    return 0;
}
