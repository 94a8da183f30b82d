/*
 * rsgpu.h — C-ABI of librsgpu.so: the MI355X (gfx950) implementation of the
 * Racing-SLAM per-frame hot path (descriptor matching -> triangulation ->
 * local-window bundle adjustment).
 *
 * This header is the whole drop-in boundary.  The reference has no FFI layer
 * (SURVEY.md §8b): its boundary is link-time, four translation units
 *   src/MapMatcher.cpp, src/Triangulation.cpp, src/Optimization.cpp,
 *   src/LocalWindow.cpp
 * behind four headers.  A maintainer replaces those four .cpp files with the
 * shims shown in INTEGRATION.md; every shim function flattens the reference's
 * pointer graph (Frame / MapPoint / Map) to the SoA arrays below and calls one
 * entry point of this header.  Each entry point cites the reference code it
 * replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no exceptions cross the ABI.
 *  - Every function returns an rs_status (0 = RS_OK).  rs_last_error() returns
 *    a human-readable message for the last failure on that context.
 *  - Pointers named d_* are DEVICE pointers (HBM of the context's GPU).
 *    Pointers named h_* are HOST pointers.  Small fixed-size parameters
 *    (poses, intrinsics, options) are host memory and are copied by value.
 *  - All work is enqueued on the context's stream (rs_context_set_stream);
 *    functions that return results in host memory synchronise that stream.
 *  - Matrices are ROW-MAJOR unless stated (Eigen's default is column-major:
 *    the shim transposes 16 floats).  A pose is the 4x4 world->camera
 *    transform exactly as Frame::pose() (src/Frame.h:46).
 *  - Descriptors are 32-byte rows (256-bit ORB, src/features/OrbFeatureExtractor.h:14-22),
 *    row i at byte offset 32*i; device descriptor arrays must be 16-byte aligned.
 *  - One context per GPU / per process rank; a context is not re-entrant
 *    (the reference's callers are single threaded, SURVEY.md §8b "Threading").
 */
#ifndef RSGPU_H
#define RSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSGPU_ABI_VERSION 3
#define RS_DESC_BYTES 32

typedef enum rs_status {
    RS_OK = 0,
    RS_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, unsorted CSR ...) */
    RS_ERR_HIP = 2,         /* a HIP runtime call failed; see rs_last_error */
    RS_ERR_NOMEM = 3,       /* workspace allocation failed */
    RS_ERR_UNSUPPORTED = 4, /* valid request outside the implemented envelope */
    RS_ERR_RCCL = 5,        /* RCCL missing or a collective failed */
    RS_ERR_NO_DEVICE = 6    /* no gfx950 device visible: there is NO CPU fallback */
} rs_status;

typedef struct rs_context rs_context;

/* ------------------------------------------------------------------ context */

int rs_abi_version(void);

/* Creates a context bound to HIP device `device_id`.  Fails with
 * RS_ERR_NO_DEVICE when no GPU is visible: the product path never falls back
 * to the CPU. */
int rs_context_create(int device_id, rs_context** out_ctx);
int rs_context_destroy(rs_context* ctx);
/* `hip_stream` is a hipStream_t (NULL = the legacy default stream). */
int rs_context_set_stream(rs_context* ctx, void* hip_stream);
int rs_context_synchronize(rs_context* ctx);
/* Stream ordering between contexts of one process, without the host: everything enqueued on `ctx` after this call
 * waits for everything enqueued so far on each of `others[0 .. n)` (one event record + one stream wait per context).
 * For hosts that run independent calls side by side — the reference's match_map, match_key_frame,
 * triangulate_tracks and match_descriptors -> triangulate_points chains share no data — on contexts of their own and
 * join them before the call that needs all their results (bench.py's pass does). */
int rs_context_wait_for(rs_context* ctx, rs_context* const* others, int n);
/* The other direction with ONE event: everything enqueued on each of `others` after this call waits for everything
 * enqueued so far on `ctx` (the fork in front of side-by-side chains). */
int rs_context_fork(rs_context* ctx, rs_context* const* others, int n);
/* Tuning knobs (integers by name).  "ba_speculative_sets": at most 1 .. 5 trust-region radii evaluated per round of
 * rs_bundle_adjust on the local-window path (0 = library default: 5; inertial solves and batches of windows: 3).  A round
 * evaluates one radius where a step is most likely accepted (first round, after two accepted steps in a row), all of them
 * while the trust region is uncalibrated (no rejection streak resolved by an accepted step yet, or the last round rejected
 * to its last set), three once it is, and never more than the iterations left.  The LM schedule, iteration count and results
 * do not depend on any of this (tests/test_gpu_parity.py), only the number of launches does.  "ba_fuse_mode" 3 needs <= 3.
 * "ba_imu_mode": rs_bundle_adjust_inertial — 0 (default) the velocity / bias blocks are eliminated around the
 * local-window reduced solve where the window allows it, 1 always the blocked solve of the full camera-side system.
 * "ba_fuse_mode": rs_bundle_adjust on a single local window (vision only, one rank) — the reduced solve and the
 * back-substitution / candidate cost of a round as ONE launch (the latter's workgroups wait for the former's hand-off
 * words inside the launch, holding a CU each): 0 (default) when no other solve of this process is in flight (lower
 * latency for one session; several sessions on one GPU get more aggregate throughput from two launches), 1 never,
 * 2 wherever possible, 3 the WHOLE round — linearisation, reduced solve, back-substitution — as one launch wherever the
 * window allows it (every workgroup resident at once: one item of landmarks per compute unit); same results.
 * "ba_band_mode": windows beyond the local-window kernels (more than 21 optimised cameras) — 0 (default) when no landmark is
 * seen by key frames more than 9 slots apart the reduced camera matrix is block-banded and is factorised by ONE launch
 * (the sparsity the reference's SPARSE_SCHUR solve exploits, src/Optimization.cpp:360): two workgroups eliminate from
 * both ends of the band towards a separator block, which a launch of its own factors; 1 always the general
 * blocked factorisation (two launches per 48 columns); 2 the banded factorisation by one workgroup from one end; same
 * results to rounding.
 * "ba_item_landmarks": landmarks per workgroup of the linearisation / Schur kernel of a single solve — 0 (default) 40
 * for windows of at most 10240 landmarks (one workgroup per compute unit), 64 beyond; or 32 / 40 / 48 / 56 / 64.
 * "ba_handoff_timeout_us": how long (1 .. 1000000, default 4000) a workgroup of that fused launch waits for a hand-off
 * word before it gives up; a solve in which that happened is re-run once as separate launches from its untouched
 * inputs (rs_ba_get_stats [4] counts them) — the caller sees the same result either way.
 * "k2_mode": rs_reproj_match — 0 (default) eight lanes per map point where the frame's KD-tree fits in LDS
 * (<= 6144 keypoints), 1 always one lane per point; the outputs are identical.
 * "ba_batch_mode": how rs_bundle_adjust_batch runs its windows — 0 (default) one launch sequence for all of them
 * where the windows allow it, 1 always the lanes. */
int rs_context_set_int(rs_context* ctx, const char* name, int value);
const char* rs_last_error(const rs_context* ctx);

/* ------------------------------------------------------------ staging pool */

/* What a drop-in translation unit does around every entry point below: upload a few freshly flattened host arrays,
 * get device scratch for the outputs, read some results back.  The pool is two grow-only bump allocators (device and
 * pinned host memory) per context; copies are asynchronous on the context stream.  Typical shim body:
 *     rs_stage_begin(ctx);                                    // recycles the pool (waits for copies still in flight)
 *     rs_stage_upload(ctx, h_desc, bytes, (void**)&d_desc);   // host -> pinned -> device, async
 *     rs_stage_alloc(ctx, out_bytes, (void**)&d_out);         // device scratch
 *     rs_xxx(ctx, d_desc, ..., d_out);                        // kernels, same stream
 *     rs_stage_download(ctx, d_out, out_bytes, h_out);        // device -> pinned, async
 *     rs_stage_sync(ctx);                                     // one synchronisation; h_out is filled on return
 * Device pointers obtained from the pool are valid until the next rs_stage_begin on the context. */
int rs_stage_begin(rs_context* ctx);
int rs_stage_alloc(rs_context* ctx, size_t bytes, void** d_out);
int rs_stage_upload(rs_context* ctx, const void* h_src, size_t bytes, void** d_out);
int rs_stage_download(rs_context* ctx, const void* d_src, size_t bytes, void* h_dst);
int rs_stage_sync(rs_context* ctx);

/* --------------------------------------------------- a4: match_descriptors */

/* Brute-force 2-nearest-neighbour search under 256-bit Hamming distance.
 * Replaces cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, knn, 2) at
 * src/MapMatcher.cpp:147-148.
 *
 * d_query [batch][nq][32] u8, d_train [batch][nt][32] u8.
 * Outputs, each [batch][nq] int32:
 *   d_idx0/d_dist0 = nearest train row and its distance,
 *   d_idx1/d_dist1 = second nearest (idx1 = -1, dist1 = -1 when nt == 1).
 * Ties: the lower train index is the nearer neighbour (OpenCV batchDistance
 * inserts with strict compares).  Bit-exact integer results.
 * nq == 0 or nt == 0 is valid: nothing is written. */
int rs_hamming_knn2(rs_context* ctx,
                    const uint8_t* d_query, int nq,
                    const uint8_t* d_train, int nt, int batch,
                    int32_t* d_idx0, int32_t* d_dist0,
                    int32_t* d_idx1, int32_t* d_dist1);

/* knnMatch + the two filters of MapMatcher::match_descriptors
 * (src/MapMatcher.cpp:150-161): keep query q iff
 *   dist0 <= max_distance                       (:152, note '>' rejects)
 *   and (nt < 2 or 4*dist0 <= 3*dist1)          (:156, MATCH_RATIO 0.75 :18)
 * Accepted matches are emitted in ascending query order (the order knn is
 * iterated in), compacted per batch item:
 *   d_match_query [batch][nq], d_match_train [batch][nq], d_match_count [batch].
 * Entries past d_match_count[b] are unspecified.
 * Any of d_idx0..d_dist1 may be NULL when the raw kNN result is not wanted. */
int rs_match_descriptors(rs_context* ctx,
                         const uint8_t* d_query, int nq,
                         const uint8_t* d_train, int nt, int batch,
                         int max_distance,
                         int32_t* d_match_query, int32_t* d_match_train,
                         int32_t* d_match_count,
                         int32_t* d_idx0, int32_t* d_dist0,
                         int32_t* d_idx1, int32_t* d_dist1);

/* ------------------------------------------- a2/a3: reprojection-gated match */

/* KD-tree over a frame's keypoints, flattened.  Mirrors KDTree2D
 * (src/KDTree.cpp:8-43): median split on x at even depth, y at odd depth,
 * mid = (start+end)/2.  Node i of the arrays is a tree node; root is node
 * `root`.  rs_kdtree_build (host) fills the three arrays (each [n]).
 * Equal-coordinate ties are resolved by (coordinate, keypoint index): the
 * reference leaves that to std::nth_element (unspecified). */
int rs_kdtree_build(const float* h_keypoints /*[n][2]*/, int n,
                    int32_t* h_node_kp /*[n] keypoint index of node*/,
                    int32_t* h_node_left /*[n] child node or -1*/,
                    int32_t* h_node_right /*[n]*/,
                    int32_t* h_root /*[1]*/);

typedef struct rs_frame_view {
    float pose[16];          /* world->camera, row-major (Frame::pose, src/Frame.h:46) */
    float fx, fy, cx, cy;    /* Camera intrinsics (src/Camera.cpp:5-13) */
    int width, height;       /* Camera::is_in_image bounds (src/Camera.cpp:34-37) */
    int n_keypoints;
    const float* d_keypoints;      /* [n][2] pixel coordinates */
    const uint8_t* d_descriptors;  /* [n][32] */
    const uint8_t* d_kp_matched;   /* [n] Frame::is_matched(index) (src/Frame.cpp:143-146) */
    const int32_t* d_kd_node_kp;   /* [n] from rs_kdtree_build */
    const int32_t* d_kd_left;      /* [n] */
    const int32_t* d_kd_right;     /* [n] */
    int kd_root;
    const void* d_kd_packed;       /* optional (NULL = not provided): the tree packed by rs_kdtree_pack for this frame,
                                      20 * n bytes, 16-byte aligned.  A frame is matched at least twice
                                      (src/Tracker.cpp:232-248): packing once saves every workgroup of both calls the
                                      dependent gathers node -> keypoint -> coordinates */
} rs_frame_view;

/* Packs the flattened KD-tree of `frame` (its d_keypoints, d_kd_* arrays) into d_packed [n] x 20 bytes for
 * rs_frame_view::d_kd_packed. */
int rs_kdtree_pack(rs_context* ctx, const rs_frame_view* frame, void* d_packed);

typedef struct rs_map_view {
    int n_points;                  /* candidate points, in MAP ORDER (src/Map.h:66) */
    const float* d_positions;      /* [P][3] MapPoint::position */
    const uint8_t* d_eligible;     /* [P] 1 = run the point.  The shim clears it for
                                      points the frame already matches (src/MapMatcher.cpp:53)
                                      and points not seen by required_observer (:169);
                                      match_for_fuse skips nulls the same way (:121-123). */
    const int32_t* d_obs_ptr;      /* [P+1] CSR: observations of point p are
                                      d_obs_ptr[p] .. d_obs_ptr[p+1]-1, in the order
                                      MapPoint::observations() is iterated */
    const int32_t* d_obs_kf;       /* [M] observing keyframe (index into d_kf_centers) */
    const int32_t* d_obs_desc;     /* [M] row of that observation's descriptor in d_desc_pool */
    const float* d_kf_centers;     /* [KF][3] Frame::camera_center (src/Frame.cpp:39-42) */
    const uint8_t* d_desc_pool;    /* [rows][32] descriptors of all keyframes */
} rs_map_view;

/* MapMatcher::match / match_for_fuse (src/MapMatcher.cpp:45-98,117-127,165-175).
 * For every eligible point: project, in-image, viewing-angle (>= 0.5) and
 * distance-range gates in f32, KD radius search r = 20 px in the reference's
 * traversal order, min Hamming over (candidate keypoint x observation) with a
 * strict '<' starting from max_distance, then per-keypoint strict-'<' argmin
 * over points in map order.
 *   replace = 0: match_map / match_key_frame (already-matched keypoints skipped)
 *   replace = 1: match_for_fuse
 * Outputs:
 *   d_point_kp   [P] best keypoint for the point or -1;  d_point_dist [P] (max_distance if none)
 *   d_prop_point [N] winning point (map order index) per keypoint or -1; d_prop_dist [N]
 *   d_match_kp / d_match_point [N] + d_match_count[1]: accepted_matches()
 *   (src/MapMatcher.cpp:34-43), ascending keypoint index.
 * Integer outputs are bit-exact against the oracle; the f32 gates follow the
 * operation order documented in oracle/reproj_match.c. */
int rs_reproj_match(rs_context* ctx, const rs_frame_view* frame,
                    const rs_map_view* map, int replace, int max_distance,
                    int32_t* d_point_kp, int32_t* d_point_dist,
                    int32_t* d_prop_point, int32_t* d_prop_dist,
                    int32_t* d_match_kp, int32_t* d_match_point,
                    int32_t* d_match_count);

/* SURVEY.md 8(e) row 2 — the same match with the MAP SHARDED over the ranks of the attached communicator (RCCL or the
 * in-process group): mp_shard holds this rank's points, point_base the map order of its first point.  The per-keypoint
 * proposal table (distance << 32 | global map order) is MIN-all-reduced before the accept step — one
 * ncclAllReduce(min, u64) of 8 N bytes — so every rank returns the unsharded result, ties included.  d_prop_point and
 * d_match_point hold GLOBAL map indices; d_point_kp / d_point_dist cover the shard.  Every rank must make the call (an
 * empty shard is valid).  Without a communicator it is rs_reproj_match with an index offset.  Worth it for maps far
 * beyond 1e5 points; smaller maps are better served by replicas. */
int rs_reproj_match_sharded(rs_context* ctx, const rs_frame_view* frame, const rs_map_view* mp_shard, int point_base,
                            int replace, int max_distance,
                            int32_t* d_point_kp, int32_t* d_point_dist,
                            int32_t* d_prop_point, int32_t* d_prop_dist,
                            int32_t* d_match_kp, int32_t* d_match_point, int32_t* d_match_count);

/* ---------------------------------------- §8(f) rank 4: the map, resident on the device */

/* MapMatcher::match walks the whole map twice per frame (src/MapMatcher.cpp:165-175); flattening the reference's
 * pointer graph for rs_reproj_match on every call costs ~100x the kernels.  rs_map keeps the flat map on the device
 * across frames; the calls below mirror, one for one, the operations the reference performs on Map / MapPoint / KeyFrame
 * (src/Map.cpp:44-124, src/MapPoint.cpp, src/Frame.cpp:80-116), each O(1) on the host; the device image is brought up
 * to date lazily, at the next use, and only for what changed (topology once per key frame; positions / key-frame
 * centres after a bundle adjustment).  Handles are small integers: point slots are never reused and ascending slot =
 * creation order = the reference's map order, which decides ties; observations of a point keep insertion order.
 * rs_frame holds a frame's keypoints, descriptors and KD-tree on the device (built and uploaded once, shared by the
 * calls of that frame); a frame promoted to a key frame hands its descriptor rows to the map device-to-device. */
typedef struct rs_map rs_map;
typedef struct rs_frame rs_frame;
int rs_map_create(rs_context* ctx, rs_map** out_map);
int rs_map_destroy(rs_map* map);
int rs_frame_create(rs_context* ctx, const float* h_keypoints /*[n][2]*/, const uint8_t* h_descriptors /*[n][32]*/, int n,
                    rs_frame** out_frame);                                        /* Frame::Frame, src/Frame.cpp:8-15 */
int rs_frame_destroy(rs_frame* frame);
int rs_map_add_keyframe(rs_map* map, const rs_frame* frame, const float h_pose[16], int* out_kf);
int rs_map_set_keyframe_pose(rs_map* map, int kf, const float h_pose[16]);      /* Frame::set_pose */
int rs_map_add_point(rs_map* map, const float h_xyz[3], int* out_point);        /* Map::add_point / create_point */
int rs_map_set_position(rs_map* map, int point, const float h_xyz[3]);         /* MapPoint::set_position */
int rs_map_remove_point(rs_map* map, int point);                                /* Map::remove_point, src/Map.cpp:63-76 */
int rs_map_add_observation(rs_map* map, int point, int kf, int keypoint);       /* Map::associate, src/Map.cpp:95-113 */
int rs_map_remove_observation(rs_map* map, int point, int kf);                 /* Map::disassociate, src/Map.cpp:115-124 */
int rs_map_counts(const rs_map* map, int h_out[4]);   /* point slots, alive points, observations, key frames */
/* MapPoint::position() of the point slots [first, first + count) from the library's mirror (removed points keep
 * their last position); what a caller copies into its own objects after rs_map_pose_graph moved the whole map. */
int rs_map_get_positions(const rs_map* map, int first, int count, float* h_xyz /*[count][3]*/);

/* MapMatcher::match_map / match_key_frame / match_for_fuse (src/MapMatcher.cpp:107-127,165-175) against the resident
 * map; results identical to rs_reproj_match on the flattened map.
 *   h_kp_matched [n] or NULL: Frame::is_matched(keypoint) (:81);  h_matched_points: slots of the points the frame already
 *   matches (:53);  required_observer_kf >= 0: match_key_frame (:169);  n_only >= 0: match_for_fuse — only the listed
 *   slots take part, and they compete for a keypoint in LIST order, as the reference's loop over its vector does
 *   (:120-126) and as the flattened path does when the shim flattens that vector in order: ONE rule on both paths
 *   (dead slots in the list are skipped like the reference's null pointers);  replace as rs_reproj_match.
 * Outputs (host, capacity n each): the accepted matches in ascending keypoint order as (keypoint, point slot). */
int rs_map_match(rs_context* ctx, rs_map* map, rs_frame* frame, const float h_pose[16], const float h_intrinsics[4],
                 int width, int height, const uint8_t* h_kp_matched, const int32_t* h_matched_points, int n_matched_points,
                 int required_observer_kf, const int32_t* h_only_points, int n_only, int replace, int max_distance,
                 int32_t* h_match_kp, int32_t* h_match_point, int* h_count);

/* (rs_map_bundle_adjust: see the optimisation section below) */

/* ------------------------------------------------------ a5-a7: triangulation */

/* triangulation::triangulate_points (src/Triangulation.cpp:37-106), batched.
 * Correspondence i uses pose table entries d_pose_idx1[i] / d_pose_idx2[i]
 * (NULL = entry 0 / entry 1: the two-frame overload :28-35; per-item indices
 * give Mapper::triangulate_tracks' pattern, src/Mapper.cpp:246-259, in one
 * launch).  P = K*[R|t] in f32 (src/Camera.cpp:47-57), DLT rows
 * x*P[2]-P[0], y*P[2]-P[1] per view, f64 one-sided Jacobi SVD, right singular
 * vector of the smallest singular value rounded to f32, then the f32 gates:
 * cheirality (:78), parallax cosine > min_parallax_cosine rejects (:83-88),
 * reprojection error > max_reprojection_error rejects (:95-100).
 *   d_xyz  [n][3] dehomogenised point of EVERY correspondence (kept or not)
 *   d_keep [n]    1 = passed all gates
 *   d_out_index [n], d_out_xyz [n][3], d_out_count[1]: the compacted
 *   TriangulatedPoint list in input order (match_index = input index).
 * n == 0 is valid (empty guard :46-48). */
int rs_triangulate(rs_context* ctx,
                   const float* d_uv1 /*[n][2]*/, const float* d_uv2 /*[n][2]*/, int n,
                   const float* d_poses /*[n_poses][16] row-major*/, int n_poses,
                   const int32_t* d_pose_idx1, const int32_t* d_pose_idx2,
                   const float h_intrinsics[4] /*fx,fy,cx,cy*/,
                   float min_parallax_cosine, float max_reprojection_error,
                   float* d_xyz, uint8_t* d_keep,
                   int32_t* d_out_index, float* d_out_xyz, int32_t* d_out_count);

/* The same function as the reference's callers see it — host vectors in, the TriangulatedPoint list out
 * (src/Triangulation.h:24-37; called with ONE correspondence per call by Mapper::triangulate_tracks, src/Mapper.cpp:253,
 * and with a handful by pose::recover_pose, src/PoseEstimation.cpp:48).  Up to 256 correspondences run as one launch of
 * one workgroup whose inputs are kernel arguments / a pinned block and whose compacted result lands in pinned memory
 * behind a completion flag: no staging copies, no stream synchronisation.  Larger n goes through the staging pool and
 * rs_triangulate.  Results are bit-identical to rs_triangulate's.
 *   h_uv1, h_uv2 [n][2];  h_out_index [n], h_out_xyz [n][3] (capacity n), *h_count = points kept. */
int rs_triangulate_host(rs_context* ctx, const float* h_uv1, const float* h_uv2, int n,
                        const float h_pose1[16], const float h_pose2[16], const float h_intrinsics[4],
                        float min_parallax_cosine, float max_reprojection_error,
                        int32_t* h_out_index, float* h_out_xyz, int* h_count);

/* triangulate_points(frame1, frame2, matches, camera) (src/Triangulation.cpp:28-35) with
 * get_matching_points (:11-26) fused and the match list left on the device:
 * correspondence i = (d_kp1[d_match_train[i]], d_kp2[d_match_query[i]]) for
 * i < min(*d_n_matches, max_matches); poses = {pose of frame1, pose of frame2}.
 * Feeds rs_match_descriptors' output straight into the triangulation with no
 * host round trip.  Outputs as rs_triangulate, indexed by match position;
 * entries at i >= *d_n_matches are not written. */
int rs_triangulate_matches(rs_context* ctx,
                           const float* d_kp1 /*[n1][2] frame1 = train side*/,
                           const float* d_kp2 /*[n2][2] frame2 = query side*/,
                           const int32_t* d_match_train, const int32_t* d_match_query,
                           const int32_t* d_n_matches /*[1] device*/, int max_matches,
                           const float* d_poses /*[2][16]*/, const float h_intrinsics[4],
                           float min_parallax_cosine, float max_reprojection_error,
                           float* d_xyz, uint8_t* d_keep,
                           int32_t* d_out_index, float* d_out_xyz, int32_t* d_out_count);

/* BASELINE.json configs[3] (64 key-frame pairs x 2k keypoints): rs_triangulate_matches for `batch` independent
 * frame pairs in one launch pair; consumes rs_match_descriptors' batched output in place.  Every array is the
 * single-pair array with a leading batch dimension: d_kp1 [batch][n1][2], d_kp2 [batch][n2][2],
 * d_match_train / d_match_query [batch][max_matches], d_n_matches [batch], d_poses [batch][2][16] (frame1, frame2),
 * d_xyz [batch][max_matches][3], d_keep / d_out_index [batch][max_matches], d_out_xyz [batch][max_matches][3],
 * d_out_count [batch].  Results per pair are those of batch single calls, bit for bit. */
int rs_triangulate_matches_batch(rs_context* ctx, int batch,
                                 const float* d_kp1, int n1, const float* d_kp2, int n2,
                                 const int32_t* d_match_train, const int32_t* d_match_query,
                                 const int32_t* d_n_matches, int max_matches,
                                 const float* d_poses, const float h_intrinsics[4],
                                 float min_parallax_cosine, float max_reprojection_error,
                                 float* d_xyz, uint8_t* d_keep,
                                 int32_t* d_out_index, float* d_out_xyz, int32_t* d_out_count);

/* §8(f) rank 1 — the body of Mapper::triangulate_tracks (reference src/Mapper.cpp:246-305):
 * per track t (track-id order, the order of the reference's std::map<TrackId, Track>,
 * src/TrackStore.h:38) with sightings CSR d_sight_ptr[t] .. d_sight_ptr[t+1] (pose index into
 * d_poses = Trajectory::pose_at(frame_index), and pixel), key-frame pixel d_track_uv[t]:
 *   - d_skip[t] != 0 or no sightings: nothing (the host-side filters of :247-250);
 *   - triangulate (first sighting, key-frame pixel) under (pose of the first sighting, d_poses[kf_pose])
 *     with gates (any_parallax_cosine = 1.0, max_reprojection_error = 4.0), :252-262;
 *   - reproject into every sighting's pose, first error > max_reprojection_error => inconsistent,
 *     :264-275 (the caller erases those tracks, :332-334);
 *   - parallax cosine and required = min(min_parallax_cosine, cos(rotation_parallax_factor * turn)),
 *     :277-288;
 * then the selection :291-304: candidates with parallax <= required in track order, and, while
 * fewer than min_new_points, the remaining candidates by parallax cosine ascending (ties: track
 * order; std::sort leaves them unspecified upstream).
 * Outputs, all device: d_status[t] 0 none / 1 candidate / 2 inconsistent; d_xyz[t][3];
 * d_parallax_cos[t]; d_required_cos[t]; d_accepted[0..counts[0]) track indices in creation order
 * (the last counts[1] of them are the top-up); d_inconsistent[0..counts[2]); d_counts[3].
 * Point creation / association (:306-330) is pointer work and stays with the caller.
 * d_required_by_pose [n_poses] or NULL: `required` depends only on the pair (pose of the first sighting, key-frame pose),
 * and it is the one quantity of this function that goes through libm (std::acos, std::cos, :282-288).  With the table —
 * rs_parallax_requirements, a HOST function using the host's libm like the reference does — required, and therefore the
 * accepted list, are bit-identical to the CPU path; with NULL the device's acosf / cosf are used (<= 3e-7 apart, which may
 * decide a track that sits exactly on its requirement differently). */
int rs_triangulate_tracks(rs_context* ctx, int n_tracks, const float* d_track_uv /*[T][2]*/,
                          const uint8_t* d_skip /*[T] or NULL*/, const int32_t* d_sight_ptr /*[T+1]*/,
                          const int32_t* d_sight_pose /*[S]*/, const float* d_sight_uv /*[S][2]*/,
                          const float* d_poses /*[n_poses][16]*/, int n_poses, int kf_pose,
                          const float h_intrinsics[4], float any_parallax_cosine,
                          float max_reprojection_error, float min_parallax_cosine,
                          float rotation_parallax_factor, int min_new_points, uint8_t* d_status,
                          float* d_xyz, float* d_parallax_cos, float* d_required_cos,
                          int32_t* d_accepted, int32_t* d_inconsistent, int32_t* d_counts /*[3]*/,
                          const float* d_required_by_pose /*[n_poses] or NULL*/);
/* Host function (no context): h_required[p] = min(min_parallax_cosine, cos(rotation_parallax_factor * turn(p))) with
 * turn(p) = acos(clamp((trace(R_kf R_p^T) - 1) / 2)) in f32, operation for operation src/Mapper.cpp:281-288. */
int rs_parallax_requirements(const float* h_poses /*[n_poses][16]*/, int n_poses, int kf_pose,
                             float min_parallax_cosine, float rotation_parallax_factor, float* h_required /*[n_poses]*/);

/* §8(f) rank 3 — the arithmetic of Mapper::cull_points (reference src/Mapper.cpp:396-431) and of
 * Slam::reprojection_error (src/Slam.cpp:302-317): for every point p with observations CSR
 * d_obs_ptr[p] .. d_obs_ptr[p+1] (pose index into d_poses, pixel) the mean over its observations of
 * |Camera::project(pose, X_p) - pixel| in f32, summed in CSR order (the reference iterates an
 * unordered_map); d_cull[p] = 1 when the point has observations and the mean exceeds max_mean_error
 * (MAX_POINT_REPROJECTION_ERROR = 3.0, src/Mapper.cpp:39); d_cull_idx[0 .. *d_cull_count) lists the culled
 * points in ascending order; d_sums = {sum of all per-observation errors (f64 sum of the f32 values),
 * number of observations}: reprojection_error() = sums[0] / sums[1].  The caller selects the local points
 * (:398-408) and removes them from the map (:426-429). */
int rs_point_errors(rs_context* ctx, int n_points, const float* d_positions /*[P][3]*/,
                    const int32_t* d_obs_ptr /*[P+1]*/, const int32_t* d_obs_pose /*[M]*/,
                    const float* d_obs_uv /*[M][2]*/, const float* d_poses /*[n_poses][16]*/, int n_poses,
                    const float h_intrinsics[4], float max_mean_error, float* d_mean_err /*[P]*/,
                    uint8_t* d_cull /*[P]*/, int32_t* d_cull_idx /*[P]*/, int32_t* d_cull_count /*[1]*/,
                    double* d_sums /*[2]*/);

/* The tail of Mapper::bundle_adjust (reference src/Mapper.cpp:366-393): single-observation points are excluded from
 * the optimisation (MIN_OBSERVATIONS_TO_OPTIMIZE, src/Optimization.cpp:98) and afterwards moved rigidly with the
 * free frame that observes them:  X' = R_after^T ((R_before X + t_before) - t_after), in f32.
 * Entry i moves point d_point_idx[i] (NULL = point i) of d_positions [.][3] with frame d_frame_idx[i];
 * d_poses_before / d_poses_after [n_frames][16] are Frame::pose() before and after rs_bundle_adjust.  A point must
 * be listed at most once (it has one observation); an entry whose frame index is outside [0, n_frames) is skipped. */
int rs_reanchor_points(rs_context* ctx, int n, const int32_t* d_point_idx, const int32_t* d_frame_idx,
                       const float* d_poses_before, const float* d_poses_after, int n_frames,
                       float* d_positions);
/* The same with the poses where the reference keeps them — on the HOST (Frame::pose(), src/Mapper.cpp:366-393):
 * h_poses_before / h_poses_after [n_frames][16] are read before the call returns.  Up to 32 frames travel as kernel
 * arguments (no upload, no extra launch: a local window has 20); more are copied to the device first. */
int rs_reanchor_points_host_poses(rs_context* ctx, int n, const int32_t* d_point_idx, const int32_t* d_frame_idx,
                                  const float* h_poses_before, const float* h_poses_after, int n_frames,
                                  float* d_positions);

/* ----------------------------------------------------- a9-a13: optimisation */

typedef enum rs_ba_termination {
    RS_BA_NO_CONVERGENCE = 0,     /* max_num_iterations reached */
    RS_BA_CONVERGENCE_FUNCTION = 1,
    RS_BA_CONVERGENCE_PARAMETER = 2,
    RS_BA_CONVERGENCE_GRADIENT = 3,
    RS_BA_CONVERGENCE_RADIUS = 4,
    RS_BA_FAILURE = 5             /* too many consecutive invalid steps / non-finite cost */
} rs_ba_termination;

/* Ceres 2.x trust-region defaults restated; solve() in the
 * reference only sets the iteration cap (src/Optimization.cpp:127-134). */
typedef struct rs_ba_options {
    int max_num_iterations;             /* BA_ITERATIONS / POSE_ITERATIONS = 10 (:118-119) */
    double huber_delta;                 /* sqrt(5.991) (:219,:311) */
    double initial_trust_region_radius; /* 1e4 */
    double max_trust_region_radius;     /* 1e16 */
    double min_trust_region_radius;     /* 1e-32 */
    double min_relative_decrease;       /* 1e-3 */
    double min_lm_diagonal;             /* 1e-6 */
    double max_lm_diagonal;             /* 1e32 */
    double function_tolerance;          /* 1e-6 */
    double gradient_tolerance;          /* 1e-10 */
    double parameter_tolerance;         /* 1e-8 */
    int max_num_consecutive_invalid_steps; /* 5 */
    int jacobi_scaling;                 /* 1 */
} rs_ba_options;

void rs_ba_default_options(rs_ba_options* opt);

typedef struct rs_ba_summary {
    int termination;        /* rs_ba_termination */
    int iterations;         /* LM iterations run (successful + unsuccessful) */
    int successful_steps;
    int usable;             /* the reference's accept rule (src/Optimization.cpp:136-141):
                               termination != FAILURE && isfinite(final) && final <= initial */
    double initial_cost;    /* 1/2 sum rho(|r|^2) at the input */
    double final_cost;
    double final_radius;
} rs_ba_summary;

/* optimization::bundle_adjust's solve (src/Optimization.cpp:269-374): Huber-robust
 * reprojection residuals (:21-72), Levenberg-Marquardt with point-block Schur
 * elimination, all in f64.
 *   d_cameras [C][6] f64 in/out: angle-axis(R_cw) then camera centre (pack_pose :144-149)
 *   h_cam_free [C] u8: FrameConfig::optimize; fixed cameras keep their residuals (:304-315)
 *   d_points  [P][3] f64 in/out: the FREE points (>= 2 observations and seen by a
 *             free frame, :287-302); points that are not free are not passed
 *   observations sorted by point (CSR): d_obs_ptr [P+1], d_obs_cam [M], d_obs_uv [M][2] f32
 * d_cameras / d_points are overwritten only when summary.usable is 1, exactly
 * like the reference writes back only on an accepted solve (:360-372).  They are
 * STREAM-ordered results: the call returns as soon as the summary (and the pinned
 * camera mirror, rs_ba_get_cameras) is on the host, possibly while the copy into
 * d_cameras / d_points is still running on the context's stream; anything that
 * reads them on that stream is ordered behind it, any other reader synchronises
 * first (rs_context_synchronize).
 * With an RCCL communicator attached (rs_comm_init_rank) the points/observations
 * are this rank's landmark shard, cameras are replicated, and the reduced
 * camera system and the cost are all-reduced every LM step. */
int rs_bundle_adjust(rs_context* ctx,
                     int n_cameras, int n_points, int n_obs,
                     double* d_cameras, const uint8_t* h_cam_free,
                     double* d_points,
                     const int32_t* d_obs_ptr, const int32_t* d_obs_cam,
                     const float* d_obs_uv,
                     const float h_intrinsics[4],
                     const rs_ba_options* options /*NULL = defaults*/,
                     rs_ba_summary* h_summary);

/* optimization::bundle_adjust (src/Optimization.cpp:269-374, vision-only) on the resident map: the window
 * (h_kfs [n_kfs] key-frame handles in FrameConfig order, h_free [n_kfs] = FrameConfig::optimize) is flattened ON THE
 * DEVICE from the resident image (three small kernels over the map's point slots), solved with rs_bundle_adjust, and on
 * a usable solve the map — device image and mirror — takes the result.  The caller gets the same for its own objects:
 * h_out_poses [n_kfs][16] (unchanged rows for fixed key frames / unusable solves) and the free points, in ascending
 * slot order, with their new positions (h_out_points / h_out_xyz, `capacity` entries; *h_n_points = how many there
 * were).  Free points: alive, >= 2 observations, matched by an optimised frame of the list (:287-302).  The flattened
 * path (host mirror / shim -> rs_bundle_adjust) lists the same set in first-seen order (the reference's, :287-302): the
 * f64 summation order inside the solve differs, so the two agree to the solver's noise (1e-9 relative), not bit for bit. */
int rs_map_bundle_adjust(rs_context* ctx, rs_map* map, const int32_t* h_kfs, const uint8_t* h_free, int n_kfs,
                         const float h_intrinsics[4], const rs_ba_options* options, rs_ba_summary* h_summary,
                         float* h_out_poses, int32_t* h_out_points, float* h_out_xyz, int capacity, int* h_n_points);

/* Throughput mode: B INDEPENDENT windows (several sessions / maps served by one GPU) in one call.  A local-window
 * solve is a chain of small dependent launches that leaves most of the 256 CUs idle.
 *   grid mode (default)  every kernel of the solve runs ONCE for all windows (grid z = window, per-window arguments in a
 *                        device table): B x 157 workgroups in the Schur kernel, B x 3 in the reduced solve, one launch
 *                        per round instead of one per window and round.  For windows of at most 64 cameras on the
 *                        local-window kernels (what a local window is); the host follows the slowest window's progress.
 *   lanes                otherwise (or "ba_batch_mode" = 1): up to `RS_BA_BATCH_LANES` child contexts (own stream, own
 *                        workspace), one host thread each, ordinary solves side by side.
 * Results per window are those of rs_bundle_adjust: the same schedule, values to summation-order noise.
 * Ordering: in both modes the windows' d_cameras / d_points are complete for anything ordered behind the call on THIS
 * context's stream (the lanes' streams are joined into it before the call returns), and rs_context_synchronize(ctx)
 * covers them.
 * h_problems[i] is the argument list of rs_bundle_adjust.  rs_ba_get_trace / _cameras refer to single solves only. */
#define RS_BA_BATCH_LANES 8
typedef struct rs_ba_problem {
    int n_cameras, n_points, n_obs;
    double* d_cameras;
    const uint8_t* h_cam_free;
    double* d_points;
    const int32_t* d_obs_ptr;
    const int32_t* d_obs_cam;
    const float* d_obs_uv;
    float intrinsics[4];
} rs_ba_problem;
int rs_bundle_adjust_batch(rs_context* ctx, int n_problems, const rs_ba_problem* h_problems,
                           const rs_ba_options* options /*NULL = defaults*/, rs_ba_summary* h_summaries /*[n_problems]*/);

/* Per-iteration record of the last rs_bundle_adjust on this context: what ceres::Solve prints with
 * minimizer_progress_to_stdout (the reference prints summary.BriefReport(), src/Optimization.cpp:135).
 * Entry i describes LM iteration i + 1.  outcome: 1 successful step, 0 rejected step, -1 invalid step
 * (linear solver failure or model_cost_change <= 0), 2 terminated by the parameter / function
 * tolerance test of this step.  Valid until the next optimisation call on the context. */
typedef struct rs_ba_iteration {
    double cost;               /* cost at x when the step was computed */
    double candidate_cost;     /* cost at x + step (0 for an invalid step) */
    double model_cost_change;
    double radius;             /* trust-region radius of this step */
    double step_norm, x_norm;  /* the two norms of the parameter-tolerance test */
    int outcome;
    int reserved0;
    double reserved1;
} rs_ba_iteration;
int rs_ba_get_trace(rs_context* ctx, rs_ba_iteration* h_out, int capacity, int* h_count);
/* The cameras as rs_bundle_adjust left them in d_cameras, from a pinned-memory mirror the last kernel of the solve
 * wrote: no device read-back, no synchronisation.  The reference keeps poses in host objects (Frame::set_pose after
 * unpack_pose, src/Optimization.cpp:363-368); the shim calls this, then rs_unpack_poses.  n_cameras must be the
 * solve's.  Valid until the next optimisation call on the context. */
int rs_ba_get_cameras(rs_context* ctx, double* h_cameras /*[n_cameras][6]*/, int n_cameras);
/* Launch accounting of the last rs_bundle_adjust: h_out[0] = rounds that did work (one K5 + K7 + K8 each),
 * [1] = rounds that relinearised (the others only re-damped after a rejected step), [2] = speculative sets
 * evaluated in total (= LM steps solved for, >= iterations), [3] = rounds enqueued by the host, [4] = solves of this
 * context (since it was created) that were run a second time as separate launches because a workgroup of the fused
 * solve + back-substitution launch timed out waiting for its hand-off word ("ba_handoff_timeout_us", default 4000:
 * a scheduling event — queue preemption, another process on the GPU — not a solver failure), [5..7] reserved (0). */
int rs_ba_get_stats(rs_context* ctx, int h_out[8]);

/* optimization::refine_pose (src/Optimization.cpp:194-267), vision-only:
 * the same residual with the points held constant, 6 unknowns.
 *   h_camera [6] f64 in/out;  d_points [n][3] f64;  d_uv [n][2] f32.
 * n == 0 returns RS_OK with summary.usable = 0 ("nothing to constrain", :227-229). */
int rs_refine_pose(rs_context* ctx, double h_camera[6],
                   const double* d_points, const float* d_uv, int n,
                   const float h_intrinsics[4],
                   const rs_ba_options* options, rs_ba_summary* h_summary);

/* ------------------------------------------- a15 / §8(f) rank 2: inertial residual blocks */

/* One IMU factor pair between two consecutive OPTIMISED frames of rs_bundle_adjust_inertial's camera list
 * (reference src/Optimization.cpp:317-346): imu::Preintegrated exactly as imu::preintegrate left it (src/Imu.h:30-40,
 * src/Imu.cpp:71-134 — that small sequential host code stays the reference's), matrices ROW-major, plus the two bias
 * random-walk densities of imu::NoiseDensity (src/Imu.h:46-48).  Each factor adds the 9-residual preintegration block
 * over (pose_i, velocity_i, bias_i, pose_j, velocity_j) whitened by L^-1 of the covariance's LLT (identity when it is not
 * positive definite, src/ImuFactor.cpp:10-17) and the 6-residual bias random walk over (bias_i, bias_j) with
 * sigma = density * sqrt(max(duration, 1e-9)) (:111-117); no loss function on either. */
typedef struct rs_imu_factor {
    int cam_i, cam_j;
    double duration;
    double rotation[9];
    double velocity[3], position[3];
    double covariance[81];
    double bias_gyro[3], bias_accel[3];     /* bias the preintegration was run with */
    double bias_jacobian[54];               /* 9 x 6 */
    double gyro_bias_sigma, accel_bias_sigma;
} rs_imu_factor;

/* optimization::bundle_adjust with InertialInput::usable() (src/Optimization.cpp:269-374 incl. :317-346).  As
 * rs_bundle_adjust, plus per-camera velocity (3) and bias (6: gyro, accel) parameter blocks for the frames the factors
 * touch: h_velocity [C][3], h_bias [C][6] in/out (host: they live in Frame::inertial(), src/Frame.h:13-16), written
 * back for the free frames on a usable solve (unpack_inertial, :363-368).  n_factors == 0 is rs_bundle_adjust.
 * The reduced camera system has 6 unknowns per free camera + 9 per inertial frame. */
int rs_bundle_adjust_inertial(rs_context* ctx,
                              int n_cameras, int n_points, int n_obs,
                              double* d_cameras, const uint8_t* h_cam_free,
                              double* d_points,
                              const int32_t* d_obs_ptr, const int32_t* d_obs_cam,
                              const float* d_obs_uv,
                              const float h_intrinsics[4],
                              double* h_velocity, double* h_bias,
                              const rs_imu_factor* h_factors, int n_factors,
                              const double h_gravity[3],
                              const rs_ba_options* options /*NULL = defaults*/,
                              rs_ba_summary* h_summary);

/* optimization::refine_pose with an InertialConstraint (src/Optimization.cpp:231-267).
 *   kind 0: none (= rs_refine_pose)
 *   kind 1: RotationPrior — h_predicted [9] row-major world->camera rotation, sigma_radians (> 0, else ignored)
 *   kind 2: InertialDelta — the previous frame's pose / velocity / bias are constant blocks (h_prev_pose [6] packed
 *           like a camera, h_prev_velocity [3], h_prev_bias [6]), h_delta its preintegration summary, h_gravity [3];
 *           this frame's velocity h_velocity [3] is a free block, written back on a usable solve (ignored when
 *           h_delta->duration <= 0, InertialDelta::enabled) */
int rs_refine_pose_inertial(rs_context* ctx, double h_camera[6],
                            const double* d_points, const float* d_uv, int n,
                            const float h_intrinsics[4], int kind,
                            const double h_predicted[9], double sigma_radians,
                            const double h_prev_pose[6], const double h_prev_velocity[3], const double h_prev_bias[6],
                            const rs_imu_factor* h_delta, const double h_gravity[3], double h_velocity[3],
                            const rs_ba_options* options, rs_ba_summary* h_summary);

/* ------------------------------------------------------------------ a14: pose graph
 * optimization::pose_graph (src/Optimization.cpp:540-639), called once per detected loop (src/Slam.cpp:258-268).
 * HOST function (no GPU work, no context): a few hundred 6-residual edges whose sparse normal equations factor in a
 * short dependent chain; the reference runs Ceres' SPARSE_NORMAL_CHOLESKY on the CPU at the same place.  The
 * data-parallel part of a loop closure, moving the map points, is rs_transform_points below.
 *   h_poses [n_kf][16]     Frame::pose() of ALL key frames in index order (Mapper::key_frames())
 *   h_loops                PoseGraphConstraint: from / to = positions in that list, relative = measured
 *                          T_from T_to^-1, row-major; entries out of range or with from == to are skipped (:590-592)
 *   four_dof, h_gravity    yaw + position only, about up = -gravity / |gravity| (falls back to SE(3) when
 *                          |gravity|^2 < 1e-6, :550-557)
 *   options                NULL = Ceres defaults with PGO_ITERATIONS = 20 (:120)
 *   h_out_poses [n_kf][16] the corrected poses, written as apply_corrected_pose (:499-504) writes them; equal to the
 *                          input when summary.usable == 0 (<=> the reference returns false and changes nothing) and
 *                          for n_kf < 3 or no loops (:546-548).  May alias h_poses.
 *   h_velocity_rotation    optional [n_kf][9] row-major R_delta of :505-509 (the caller rotates
 *                          InertialState::velocity by it); identity where nothing moved
 *   h_trace                optional per-iteration record, as rs_ba_get_trace; *h_trace_count = iterations recorded */
typedef struct rs_pose_graph_edge {
    int32_t from, to;
    double relative[16];
} rs_pose_graph_edge;
int rs_pose_graph(int n_kf, const float* h_poses, const rs_pose_graph_edge* h_loops, int n_loops, int four_dof,
                  const double h_gravity[3], const rs_ba_options* options, float* h_out_poses,
                  float* h_velocity_rotation, rs_ba_summary* h_summary, rs_ba_iteration* h_trace, int trace_capacity,
                  int* h_trace_count);
/* pose_relative (:494-497): T_from (widened) * inverse(T_to) (f32 cofactor inverse, widened) — what the sequential
 * edges measure; exposed so that a caller can build loop constraints the same way. */
void rs_pose_relative(const float h_from[16], const float h_to[16], double h_relative[16]);
/* transform_points (:512-536) on the device: every point with observations moves rigidly with its owner, the
 * observing key frame of smallest index:  X' = R_after^T ((R_before X + t_before) - t_after), f32.
 * Observation CSR as in rs_map_view (d_obs_ptr [n_points + 1], d_obs_kf [.] = position in the key-frame list);
 * d_poses_before / d_poses_after [n_kf][16], 16-byte aligned. */
int rs_transform_points(rs_context* ctx, int n_points, const int32_t* d_obs_ptr, const int32_t* d_obs_kf,
                        const float* d_poses_before, const float* d_poses_after, int n_kf, float* d_positions);
/* pose_graph on the resident map: poses from the mirror (every key frame of the map, in handle order), rs_pose_graph,
 * and on a usable solve the key frames take the corrected poses and K14 moves the points on the device (the mirror
 * follows).  h_out_poses [n_kf][16] / h_velocity_rotation [n_kf][9] (optional) are for the caller's own objects;
 * returns summary.usable == 0 and changes nothing exactly when the reference returns false. */
int rs_map_pose_graph(rs_context* ctx, rs_map* map, const rs_pose_graph_edge* h_loops, int n_loops, int four_dof,
                      const double h_gravity[3], const rs_ba_options* options, float* h_out_poses,
                      float* h_velocity_rotation, rs_ba_summary* h_summary);

/* pack_pose / unpack_pose (src/Optimization.cpp:144-159, a10).  Host only:
 * R -> angle-axis in f32 through a quaternion (ceres::RotationMatrixToAngleAxis<float>),
 * centre = -R^T t (src/Frame.cpp:39-42), widened to f64; and back. */
void rs_pack_pose(const float h_pose[16], double h_camera[6]);
void rs_unpack_pose(const double h_camera[6], float h_pose[16]);
/* the loops around them (src/Optimization.cpp:273-282, 363-368); h_mask (NULL = all) selects the frames written */
void rs_pack_poses(const float* h_poses /*[n][16]*/, int n, double* h_cameras /*[n][6]*/);
void rs_unpack_poses(const double* h_cameras /*[n][6]*/, int n, const uint8_t* h_mask /*[n] or NULL*/, float* h_poses /*[n][16]*/);

/* ------------------------------------------------------- a8: local window */

/* optimization::build_local_window (src/LocalWindow.cpp:10-52).  Host only.
 * Keyframes are 0..n_key_frames-1 in Mapper order; new_frame is the index of
 * the new frame in that list, or -1 when it is not (yet) a keyframe.
 * The covisibility input is CSR: frame f matches points
 * h_frame_pt[h_frame_ptr[f] .. h_frame_ptr[f+1]-1]; rows 0..n_key_frames-1 are
 * the key frames (h_frame_ptr has n_key_frames + 1 entries) and, only when
 * new_frame == -1, row n_key_frames is the new frame (n_key_frames + 2
 * entries); point p is observed by
 * keyframes h_pt_obs[h_pt_ptr[p] .. h_pt_ptr[p+1]-1].
 * Output (capacity n_key_frames+1): h_out_frame[i] (n_key_frames = the new
 * non-keyframe), h_out_optimize[i]; *h_out_count entries. */
int rs_build_local_window(int n_key_frames, int new_frame, int window_size, int fix_oldest,
                          const int32_t* h_frame_ptr, const int32_t* h_frame_pt,
                          const int32_t* h_pt_ptr, const int32_t* h_pt_obs,
                          int32_t* h_out_frame, uint8_t* h_out_optimize, int32_t* h_out_count);

/* ------------------------------------------------------------- multi-GPU */

#define RS_COMM_ID_BYTES 128
/* One process per GPU.  Rank 0 calls rs_comm_get_unique_id, the host program
 * distributes the bytes (bench.py: torch.distributed broadcast), every rank
 * calls rs_comm_init_rank.  Only rs_bundle_adjust communicates (sum all-reduce
 * of the reduced camera system + cost over RCCL/xGMI); matching and
 * triangulation shard with no collective. */
int rs_comm_get_unique_id(uint8_t h_id[RS_COMM_ID_BYTES]);
int rs_comm_init_rank(rs_context* ctx, const uint8_t h_id[RS_COMM_ID_BYTES], int n_ranks, int rank);
int rs_comm_destroy(rs_context* ctx);
/* The same exchange step WITHOUT RCCL for n <= 8 contexts of one process (one host thread each, each with its own
 * stream, on one device or on peer-accessible devices): context i becomes rank i of an in-process group whose
 * all-reduce is a deterministic on-device sum in rank order.  Every member must then make the same sequence of
 * rs_bundle_adjust calls, each from its own thread.  Used to run landmark shards side by side on one GPU and to
 * test the N > 1 path on a one-GPU box.  rs_comm_destroy on every member releases the group.
 * A group in which an exchange step failed (a member's HIP error, or a member that did not arrive within 30 s) stays
 * failed: every later exchange step of every member returns RS_ERR_HIP at once ("... failed in an earlier exchange
 * step"); destroy the group on every member and create it again. */
int rs_comm_init_local(rs_context** ctxs, int n);
/* Evidence of what the exchange step runs over: *h_ranks = number of ranks of the attached communicator as RCCL
 * itself reports it (ncclCommCount) or the size of the in-process group; *h_kind = 0 none, 1 RCCL, 2 in-process. */
int rs_comm_count(rs_context* ctx, int* h_ranks, int* h_kind);

/* ------------------------------------------------------------- profiling */

/* Per-kernel HIP-event timing on the context's stream (the hipEvent analogue
 * of the reference's time_it(), src/Helpers.h:8-25).  Between begin and end
 * every kernel the library launches is bracketed by events; end synchronises
 * and returns the totals.  Kernel names are stable identifiers (K1..K9). */
#define RS_PROF_MAX 32
typedef struct rs_prof_entry {
    char name[32];
    int launches;
    double total_ms;
} rs_prof_entry;
int rs_prof_begin(rs_context* ctx);
int rs_prof_end(rs_context* ctx, rs_prof_entry* h_entries /*[RS_PROF_MAX]*/, int* h_count);
/* Diagnostic: shader-cycle totals per in-kernel phase of the last rs_bundle_adjust
 * call (thread 0 of workgroup 0 stamps s_memtime at phase boundaries; slots 0-6
 * K7, 8-14 K5).  n <= 64. */
int rs_prof_counters(rs_context* ctx, uint64_t* h_out, int n);
/* Launch latency of this device as the library's own launches see it ("state the measured empty-launch
 * latency next to the numbers"): n back-to-back launches of an empty one-workgroup kernel on the context stream
 * between two HIP events; *h_us = microseconds per launch. */
int rs_prof_empty_launch(rs_context* ctx, int n, double* h_us);

#ifdef __cplusplus
}
#endif
#endif /* RSGPU_H */
