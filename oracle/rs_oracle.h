/*
 * rs_oracle.h — CPU restatement (plain C) of the Racing-SLAM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (GregVS/Racing-SLAM) ships no tests, golden
 * vectors or fixtures for this path and cannot be built
 * here (OpenCV, Eigen and Ceres are absent from the image).  Each function
 * restates the reference file:line it cites plus the published algorithm of
 * the third-party call underneath (OpenCV 4.x BFMatcher / triangulatePoints,
 * Ceres 2.x trust-region LM + Schur; pinned only by vcpkg baseline
 * 4b6c50d962cc20aaa3ef457f8ba683b586263cfb).  The restatement is validated in
 * tests/ against independent numpy/scipy formulations.
 *
 * All pointers are host pointers.  Layouts are identical to include/rsgpu.h so
 * one set of arrays drives both the oracle and the GPU library.
 */
#ifndef RS_ORACLE_H
#define RS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* -- hamming.c ---------------------------------------------------------- */
int orc_hamming_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt,
                     int32_t* idx0, int32_t* dist0, int32_t* idx1, int32_t* dist1);
int orc_match_descriptors(const uint8_t* query, int nq, const uint8_t* train, int nt,
                          int max_distance, int32_t* match_query, int32_t* match_train,
                          int32_t* match_count);

/* -- kdtree.c ----------------------------------------------------------- */
int orc_kdtree_build(const float* keypoints, int n, int32_t* node_kp, int32_t* node_left,
                     int32_t* node_right, int32_t* root);
/* radius search in the reference's traversal order; returns count (<= cap) */
int orc_kdtree_radius(const float* keypoints, const int32_t* node_kp, const int32_t* node_left,
                      const int32_t* node_right, int root, float x, float y, float radius,
                      int32_t* out, int cap);

/* -- reproj_match.c ----------------------------------------------------- */
typedef struct orc_frame_view {
    float pose[16];
    float fx, fy, cx, cy;
    int width, height;
    int n_keypoints;
    const float* keypoints;
    const uint8_t* descriptors;
    const uint8_t* kp_matched;
    const int32_t* kd_node_kp;
    const int32_t* kd_left;
    const int32_t* kd_right;
    int kd_root;
} orc_frame_view;

typedef struct orc_map_view {
    int n_points;
    const float* positions;
    const uint8_t* eligible;
    const int32_t* obs_ptr;
    const int32_t* obs_kf;
    const int32_t* obs_desc;
    const float* kf_centers;
    const uint8_t* desc_pool;
} orc_map_view;

int orc_reproj_match(const orc_frame_view* frame, const orc_map_view* map, int replace,
                     int max_distance, int32_t* point_kp, int32_t* point_dist,
                     int32_t* prop_point, int32_t* prop_dist, int32_t* match_kp,
                     int32_t* match_point, int32_t* match_count);

/* -- triangulate.c ------------------------------------------------------ */
int orc_triangulate(const float* uv1, const float* uv2, int n, const float* poses, int n_poses,
                    const int32_t* pose_idx1, const int32_t* pose_idx2,
                    const float intrinsics[4], float min_parallax_cosine,
                    float max_reprojection_error, float* xyz, uint8_t* keep,
                    int32_t* out_index, float* out_xyz, int32_t* out_count);
/* -- tracks.c: the body of Mapper::triangulate_tracks (src/Mapper.cpp:246-305) ------------- */
int orc_triangulate_tracks(int n_tracks, const float* track_uv, const uint8_t* skip, const int32_t* sight_ptr,
                           const int32_t* sight_pose, const float* sight_uv, const float* poses, int n_poses,
                           int kf_pose, const float intrinsics[4], float any_parallax_cosine,
                           float max_reprojection_error, float min_parallax_cosine,
                           float rotation_parallax_factor, int min_new_points, uint8_t* status, float* xyz,
                           float* parallax_cos, float* required_cos, int32_t* accepted, int32_t* n_accepted,
                           int32_t* n_topped_up, int32_t* inconsistent, int32_t* n_inconsistent);
int orc_point_errors(int n_points, const float* positions, const int32_t* obs_ptr, const int32_t* obs_pose,
                     const float* obs_uv, const float* poses, int n_poses, const float intrinsics[4],
                     float max_mean_error, float* mean_err, uint8_t* cull, int32_t* cull_idx, int32_t* cull_count,
                     double sums[2]);
int orc_reanchor_points(int n, const int32_t* point_idx, const int32_t* frame_idx, const float* before,
                        const float* after, float* positions);
/* transform_points of the pose graph, reference src/Optimization.cpp:512-536: every point with observations moves rigidly
 * with its OWNER = the observing key frame of smallest index (obs_kf holds positions in the key-frame list, which is in
 * index order; an owner < 0, i.e. not in the list, leaves the point alone). */
int orc_transform_points(int n_points, const int32_t* obs_ptr, const int32_t* obs_kf, const float* before,
                         const float* after, float* positions);
/* 4x4 f64 one-sided Jacobi SVD null vector (exposed for tests): v = right
 * singular vector of the smallest singular value of row-major A. */
void orc_null_vector4(const double A[16], double v[4], double sigma[4]);

/* -- rotation.c --------------------------------------------------------- */
void orc_pack_pose(const float pose[16], double camera[6]);
void orc_unpack_pose(const double camera[6], float pose[16]);
void orc_angle_axis_rotate_point(const double aa[3], const double pt[3], double out[3]);

/* -- ba.c --------------------------------------------------------------- */
typedef struct orc_ba_options {
    int max_num_iterations;
    double huber_delta;
    double initial_trust_region_radius;
    double max_trust_region_radius;
    double min_trust_region_radius;
    double min_relative_decrease;
    double min_lm_diagonal;
    double max_lm_diagonal;
    double function_tolerance;
    double gradient_tolerance;
    double parameter_tolerance;
    int max_num_consecutive_invalid_steps;
    int jacobi_scaling;
} orc_ba_options;

typedef struct orc_ba_summary {
    int termination;
    int iterations;
    int successful_steps;
    int usable;
    double initial_cost;
    double final_cost;
    double final_radius;
} orc_ba_summary;

void orc_ba_default_options(orc_ba_options* opt);

/* Per-iteration record of the trust-region loop (what Ceres prints with
 * minimizer_progress_to_stdout): one entry per LM iteration, in order.
 *   outcome 1 = successful step, 0 = rejected (rho <= min_relative_decrease),
 *          -1 = invalid step (solver failure or model_cost_change <= 0),
 *           2 = the loop terminated on this step's parameter / function tolerance test. */
typedef struct orc_ba_iteration {
    double cost;               /* cost at x when the step was computed */
    double candidate_cost;     /* cost at x + step (0 for an invalid step) */
    double model_cost_change;
    double radius;             /* trust-region radius the step was computed with */
    double step_norm, x_norm;  /* as in the parameter-tolerance test */
    int outcome;
    int pad;
} orc_ba_iteration;
/* The next orc_bundle_adjust / orc_refine_pose calls on this thread fill buf[0 .. *count);
 * pass NULL to switch tracing off. */
void orc_ba_set_trace(orc_ba_iteration* buf, int capacity, int* count);

/* residual + autodiff-equivalent Jacobians of one observation (forward-mode
 * jets through the reference functor).  jc [2][6], jp [2][3], row-major. */
void orc_reprojection(const double cam[6], const double pt[3], const float uv[2],
                      const float intrinsics[4], double r[2], double jc[12], double jp[6]);

int orc_bundle_adjust(int n_cameras, int n_points, int n_obs, double* cameras,
                      const uint8_t* cam_free, double* points, const int32_t* obs_ptr,
                      const int32_t* obs_cam, const float* obs_uv, const float intrinsics[4],
                      const orc_ba_options* options, orc_ba_summary* summary);

/* -- imu.c: the inertial residual blocks (src/ImuFactor.cpp:10-118, src/Optimization.cpp:74-95) ------------- */
/* One IMU factor pair between two consecutive optimised frames: imu::Preintegrated (src/Imu.h:30-40) as
 * imu::preintegrate left it (ROW-major matrices), plus the two bias random-walk densities of imu::NoiseDensity. */
typedef struct orc_imu_factor {
    int cam_i, cam_j;
    double duration;
    double rotation[9];
    double velocity[3], position[3];
    double covariance[81];
    double bias_gyro[3], bias_accel[3];
    double bias_jacobian[54];
    double gyro_bias_sigma, accel_bias_sigma;
} orc_imu_factor;
void orc_imu_whitener(const double cov[81], double W[81]);
void orc_imu_preintegration(const orc_imu_factor* f, const double gravity[3], const double pose_i[6], const double vel_i[3],
                            const double bias_i[6], const double pose_j[6], const double vel_j[3], double r[9], double J[216]);
void orc_imu_bias_walk(const orc_imu_factor* f, const double bias_i[6], const double bias_j[6], double r[6], double J[72]);
void orc_rotation_prior(const double predicted[9], double sigma, const double pose[6], double r[3], double J[18]);
int orc_bundle_adjust_inertial(int n_cameras, int n_points, int n_obs, double* cameras, const uint8_t* cam_free,
                               double* points, const int32_t* obs_ptr, const int32_t* obs_cam, const float* obs_uv,
                               const float intrinsics[4], double* velocity, double* bias, const orc_imu_factor* factors,
                               int n_factors, const double gravity[3], const orc_ba_options* options,
                               orc_ba_summary* summary);
int orc_refine_pose_inertial(double camera[6], const double* points, const float* uv, int n, const float intrinsics[4],
                             int kind, const double predicted[9], double sigma, const double prev_pose[6],
                             const double prev_velocity[3], const double prev_bias[6], const orc_imu_factor* delta,
                             const double gravity[3], double velocity[3], const orc_ba_options* options,
                             orc_ba_summary* summary);

/* -- pose_graph.c: optimization::pose_graph, reference src/Optimization.cpp:376-639 -------------------------------
 * poses [n][16] f32 row-major world->camera of ALL key frames in index order; loops: from / to = positions in that
 * list, relative = measured from * to^-1 (row-major 4x4 f64).  out_poses = input when the result is not usable
 * (summary->usable == 0 <=> the reference returns false).  options NULL = Ceres defaults with 20 iterations. */
typedef struct orc_pg_edge {
    int32_t from, to;
    double relative[16];
} orc_pg_edge;
int orc_pose_graph(int n, const float* poses, const orc_pg_edge* loops, int n_loops, int four_dof, const double gravity[3],
                   const orc_ba_options* options, float* out_poses, orc_ba_summary* summary);
/* pose_relative, :494-497: from (widened) * inverse(to) (f32 cofactor inverse, widened); row-major 4x4 */
void orc_pose_relative(const float from[16], const float to[16], double rel[16]);
/* residual [6] and Jacobian [6][12] (SE3: from 0-5 | to 6-11; 4-DoF: from 0-3 | to 4-7, rest 0) of one edge, for tests */
void orc_pose_graph_edge(int four_dof, const double* x_from, const double* x_to, const double R0_from[9], const double R0_to[9],
                         const double up[3], const double relative[16], int loop, double r[6], double J[72]);

/* One linearisation at the given state (no update), for tests and for the
 * multi-GPU sharding tests: fills the UNDAMPED normal equations
 *   U [C][6][6], gc [C][6], V [P][3][3], gp [P][3] and the robust cost. */
int orc_ba_linearize(int n_cameras, int n_points, const double* cameras, const double* points,
                     const int32_t* obs_ptr, const int32_t* obs_cam, const float* obs_uv,
                     const float intrinsics[4], double huber_delta, double* U, double* gc,
                     double* V, double* gp, double* cost);

int orc_refine_pose(double camera[6], const double* points, const float* uv, int n,
                    const float intrinsics[4], const orc_ba_options* options,
                    orc_ba_summary* summary);

/* -- local_window.c ----------------------------------------------------- */
int orc_build_local_window(int n_key_frames, int new_frame, int window_size, int fix_oldest,
                           const int32_t* frame_ptr, const int32_t* frame_pt,
                           const int32_t* pt_ptr, const int32_t* pt_obs, int32_t* out_frame,
                           uint8_t* out_optimize, int32_t* out_count);

#ifdef __cplusplus
}
#endif
#endif
