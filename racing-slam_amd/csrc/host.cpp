// host.cpp — the host-only entry points of librsgpu.so: pieces of the hot path
// that are pointer-set logic or O(cameras) scalar math and stay on the CPU
// (SURVEY.md §8 a8, a10): pose packing, the keypoint KD-tree build and
// build_local_window.  No GPU work and no dependency on the oracle.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "../../include/rsgpu.h"

namespace {

constexpr double kDblEps = 2.220446049250313e-16;

// ceres::RotationMatrixToQuaternion<float> on the 3x3 block of a row-major 4x4
void matrix_to_quat(const float* T, float q[4])
{
    auto R = [&](int i, int j) { return T[4 * i + j]; };
    const float trace = R(0, 0) + R(1, 1) + R(2, 2);
    if (trace >= 0.0f) {
        float t = sqrtf(trace + 1.0f);
        q[0] = 0.5f * t;
        t = 0.5f / t;
        q[1] = (R(2, 1) - R(1, 2)) * t;
        q[2] = (R(0, 2) - R(2, 0)) * t;
        q[3] = (R(1, 0) - R(0, 1)) * t;
        return;
    }
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    float t = sqrtf(R(i, i) - R(j, j) - R(k, k) + 1.0f);
    q[i + 1] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (R(k, j) - R(j, k)) * t;
    q[j + 1] = (R(j, i) + R(i, j)) * t;
    q[k + 1] = (R(k, i) + R(i, k)) * t;
}

// ceres::QuaternionToAngleAxis<float>
void quat_to_angle_axis(const float q[4], float aa[3])
{
    const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    float k = 2.0f;
    if (s2 > 0.0f) {
        const float s = sqrtf(s2), c = q[0];
        const float two_theta = 2.0f * (c < 0.0f ? atan2f(-s, -c) : atan2f(s, c));
        k = two_theta / s;
    }
    aa[0] = q[1] * k;
    aa[1] = q[2] * k;
    aa[2] = q[3] * k;
}

}  // namespace

// pack_pose, reference src/Optimization.cpp:144-149 (matrix_to_rodrigues :107-112,
// Frame::camera_center src/Frame.cpp:39-42)
extern "C" void rs_pack_pose(const float T[16], double cam[6])
{
    float q[4], aa[3];
    matrix_to_quat(T, q);
    quat_to_angle_axis(q, aa);
    for (int i = 0; i < 3; i++) {
        const float centre = (-T[i] * T[3] + -T[4 + i] * T[7]) + -T[8 + i] * T[11];
        cam[i] = (double)aa[i];
        cam[3 + i] = (double)centre;
    }
}

// The loops around pack_pose / unpack_pose in bundle_adjust (reference src/Optimization.cpp:273-282, 363-368):
// `mask` (may be NULL) selects the frames, like FrameConfig::optimize on write-back.
extern "C" void rs_unpack_pose(const double cam[6], float T[16]);
extern "C" void rs_pack_poses(const float* h_poses, int n, double* h_cameras)
{
    for (int i = 0; i < n; i++) rs_pack_pose(h_poses + 16 * (size_t)i, h_cameras + 6 * (size_t)i);
}
extern "C" void rs_unpack_poses(const double* h_cameras, int n, const uint8_t* h_mask, float* h_poses)
{
    for (int i = 0; i < n; i++)
        if (!h_mask || h_mask[i]) rs_unpack_pose(h_cameras + 6 * (size_t)i, h_poses + 16 * (size_t)i);
}

// unpack_pose, reference src/Optimization.cpp:151-159 (rodrigues_to_matrix :100-105)
extern "C" void rs_unpack_pose(const double cam[6], float T[16])
{
    const float ax = (float)cam[0], ay = (float)cam[1], az = (float)cam[2];
    const float c[3] = {(float)cam[3], (float)cam[4], (float)cam[5]};
    float R[3][3];
    const float theta2 = ax * ax + ay * ay + az * az;
    if (theta2 > (float)kDblEps) {
        const float theta = sqrtf(theta2);
        const float wx = ax / theta, wy = ay / theta, wz = az / theta;
        const float ct = cosf(theta), st = sinf(theta);
        R[0][0] = ct + wx * wx * (1.0f - ct);
        R[1][0] = wz * st + wx * wy * (1.0f - ct);
        R[2][0] = -wy * st + wx * wz * (1.0f - ct);
        R[0][1] = wx * wy * (1.0f - ct) - wz * st;
        R[1][1] = ct + wy * wy * (1.0f - ct);
        R[2][1] = wx * st + wy * wz * (1.0f - ct);
        R[0][2] = wy * st + wx * wz * (1.0f - ct);
        R[1][2] = -wx * st + wy * wz * (1.0f - ct);
        R[2][2] = ct + wz * wz * (1.0f - ct);
    } else {
        R[0][0] = 1.0f; R[1][0] = az; R[2][0] = -ay;
        R[0][1] = -az; R[1][1] = 1.0f; R[2][1] = ax;
        R[0][2] = ay; R[1][2] = -ax; R[2][2] = 1.0f;
    }
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[4 * i + j] = R[i][j];
        T[4 * i + 3] = (-R[i][0] * c[0] + -R[i][1] * c[1]) + -R[i][2] * c[2];
    }
    T[12] = T[13] = T[14] = 0.0f;
    T[15] = 1.0f;
}

// KDTree2D::build, reference src/KDTree.cpp:8-43, flattened: node id = position
// of its keypoint in the final permutation.  Iterative (explicit work list).
extern "C" int rs_kdtree_build(const float* kp, int n, int32_t* node_kp, int32_t* node_left,
                               int32_t* node_right, int32_t* root)
{
    if (n < 0 || !root) return RS_ERR_INVALID;
    *root = -1;
    if (n == 0) return RS_OK;
    if (!kp || !node_kp || !node_left || !node_right) return RS_ERR_INVALID;
    std::iota(node_kp, node_kp + n, 0);
    struct Item { int start, end, depth; int32_t* link; };
    std::vector<Item> work;
    work.push_back({0, n, 0, root});
    while (!work.empty()) {
        const Item it = work.back();
        work.pop_back();
        if (it.start >= it.end) { *it.link = -1; continue; }
        const int axis = it.depth % 2;
        const int mid = (it.start + it.end) / 2;
        // nth_element semantics with a total order: (coordinate, keypoint index)
        std::nth_element(node_kp + it.start, node_kp + mid, node_kp + it.end, [&](int32_t a, int32_t b) {
            const float va = kp[2 * a + axis], vb = kp[2 * b + axis];
            return va < vb || (va == vb && a < b);
        });
        *it.link = mid;
        work.push_back({it.start, mid, it.depth + 1, node_left + mid});
        work.push_back({mid + 1, it.end, it.depth + 1, node_right + mid});
    }
    return RS_OK;
}

// build_local_window, reference src/LocalWindow.cpp:10-52
extern "C" int rs_build_local_window(int n_key_frames, int new_frame, int window_size, int fix_oldest,
                                     const int32_t* frame_ptr, const int32_t* frame_pt, const int32_t* pt_ptr,
                                     const int32_t* pt_obs, int32_t* out_frame, uint8_t* out_optimize,
                                     int32_t* out_count)
{
    if (n_key_frames < 0 || window_size < 0 || new_frame >= n_key_frames || !out_count) return RS_ERR_INVALID;
    if (!frame_ptr || !pt_ptr || !out_frame || !out_optimize) return RS_ERR_INVALID;
    const int n = n_key_frames;
    const int self = new_frame >= 0 ? new_frame : n;
    const int first_optimized = n > window_size ? n - window_size : 2;            // :15
    enum : uint8_t { kOutside = 0, kWindow = 1, kAnchor = 2 };
    std::vector<uint8_t> role((size_t)n + 1, kOutside);
    role[self] = kWindow;                                                          // :16
    for (int i = first_optimized; i < n; i++) role[i] = kWindow;                  // :17-19
    // :21-30.  The reference walks (window frame, matched point, observer) triples; the result is a SET of observers,
    // so every point needs to be visited once only, and the walk can stop when no key frame is left outside.
    int outside = 0;
    for (int i = 0; i < n; i++) outside += role[i] == kOutside ? 1 : 0;
    if (outside > 0) {
        const int n_rows = new_frame >= 0 ? n : n + 1;        // row n of the CSR exists only for a new non-key frame
        int n_pts = 0;
        for (int f = 0; f < n_rows; f++)
            for (int a = frame_ptr[f]; a < frame_ptr[f + 1]; a++) n_pts = frame_pt[a] >= n_pts ? frame_pt[a] + 1 : n_pts;
        std::vector<uint8_t> seen((size_t)n_pts, 0);
        for (int f = 0; f < n_rows && outside > 0; f++) {
            if (role[f] != kWindow) continue;
            for (int a = frame_ptr[f]; a < frame_ptr[f + 1] && outside > 0; a++) {
                const int p = frame_pt[a];
                if (seen[p]) continue;
                seen[p] = 1;
                for (int o = pt_ptr[p]; o < pt_ptr[p + 1]; o++) {
                    uint8_t& r = role[pt_obs[o]];
                    if (r == kOutside) { r = kAnchor; outside--; }
                }
            }
        }
    }
    int count = 0;
    bool included = false;
    for (int i = 0; i < n; i++) {                                                  // :35-47
        const bool fixed = i < 2 || (fix_oldest && i == first_optimized);
        if (role[i] == kWindow) {
            out_frame[count] = i;
            out_optimize[count++] = fixed ? 0 : 1;
            included = included || i == self;
        } else if (fixed || role[i] == kAnchor) {
            out_frame[count] = i;
            out_optimize[count++] = 0;
        }
    }
    if (!included) {                                                               // :48-50
        out_frame[count] = self;
        out_optimize[count++] = 1;
    }
    *out_count = count;
    return RS_OK;
}

// The rotation-dependent parallax requirement of Mapper::triangulate_tracks (reference src/Mapper.cpp:281-288) per pose of a
// first sighting: the one quantity of that function that goes through libm.  Computed here with the HOST's acosf / cosf —
// what the reference itself calls — in the oracle's operation order (f32, dot products as (a0 b0 + a1 b1) + a2 b2, no FMA
// contraction: this file is built with -ffp-contract=off), so that K6 with this table selects exactly the CPU path's tracks.
extern "C" int rs_parallax_requirements(const float* h_poses, int n_poses, int kf_pose, float min_parallax_cosine,
                                        float rotation_parallax_factor, float* h_required)
{
    if (n_poses < 0 || (n_poses > 0 && (!h_poses || !h_required))) return RS_ERR_INVALID;
    if (n_poses == 0) return RS_OK;
    if (kf_pose < 0 || kf_pose >= n_poses) return RS_ERR_INVALID;
    const float* Tk = h_poses + 16 * (size_t)kf_pose;
    for (int p = 0; p < n_poses; p++) {
        const float* Tf = h_poses + 16 * (size_t)p;
        float tr[3];
        for (int i = 0; i < 3; i++) tr[i] = (Tk[4 * i] * Tf[4 * i] + Tk[4 * i + 1] * Tf[4 * i + 1]) + Tk[4 * i + 2] * Tf[4 * i + 2];
        const float trace = (tr[0] + tr[1]) + tr[2];
        float cosine = (trace - 1.0f) / 2.0f;
        cosine = cosine < -1.0f ? -1.0f : cosine;
        cosine = cosine > 1.0f ? 1.0f : cosine;
        const float turned = acosf(cosine);
        const float need = cosf(rotation_parallax_factor * turned);
        h_required[p] = min_parallax_cosine < need ? min_parallax_cosine : need;
    }
    return RS_OK;
}
